"""Fused PV-net (csrc/az_net.hip) against the plain PyTorch fp32 forward of the same weights.

Floating point, so a tolerance applies (north_star: "within stochastic-sampling tolerance"): the kernel keeps
fp16 weights/activations with fp32 accumulation and an fp32 residual stream.  Bar used here: max |dprior| <= 4e-3
and max |dvalue| <= 8e-3 against torch fp32 on CPU; the CPU-only tests check the host-side packing with a numpy
emulation of the kernel's data movement.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from alphazero_openspiel_amd import fusednet, games
from alphazero_openspiel_amd.network import Net, load_npz_checkpoint, state_to_board

P_TOL, V_TOL = 4e-3, 8e-3


def _random_boards(game, n, seed):
    rng = np.random.RandomState(seed)
    out = []
    s = game.new_initial_state()
    while len(out) < n:
        if s.is_terminal():
            s = game.new_initial_state()
        out.append(state_to_board(s, game.information_state_normalized_vector_shape()))
        la = s.legal_actions()
        s.apply_action(la[rng.randint(len(la))])
    return np.array(out, dtype=np.float32)


def _nets():
    c4 = games.load_game("connect_four")
    bt6 = games.load_game("breakthrough(rows=6,columns=6)")
    bt8 = games.load_game("breakthrough(rows=8,columns=8)")
    torch.manual_seed(3)
    return {
        "c4_ckpt": (c4, load_npz_checkpoint(os.path.join(GOLDEN, "checkpoint_connect_four.npz"), [3, 6, 7], 7)),
        "bt6_ckpt": (bt6, load_npz_checkpoint(os.path.join(GOLDEN, "checkpoint_breakthrough6.npz"), [3, 6, 6], 432)),
        "c4_10block": (c4, Net([3, 6, 7], 7, n_blocks=10, n_filters=50).eval()),
        "bt8_2block": (bt8, Net([3, 8, 8], 768, n_blocks=2, n_filters=50).eval()),
        # the nets of BASELINE configs 3 and 5 at their full batch sizes (split logits / softmax head kernels)
        "bt6_10block": (bt6, Net([3, 6, 6], 432, n_blocks=10, n_filters=50).eval()),
        "bt8_20block": (bt8, Net([3, 8, 8], 768, n_blocks=20, n_filters=50).eval()),
        "bt5x4_3block": (games.load_game("breakthrough(rows=5,columns=4)"),
                         Net([3, 5, 4], 240, n_blocks=3, n_filters=32).eval()),
        # > 50 filters: output-channel tile 3 is streamed in full (the <= 50 case streams 2 of its 16 rows)
        "c4_56f_2block": (c4, Net([3, 6, 7], 7, n_blocks=2, n_filters=56).eval()),
        # 25 cells: 50 k-steps of fc1, not a whole number of the logits kernel's 4-k-step weight chunks
        "bt5x5_2block": (games.load_game("breakthrough(rows=5,columns=5)"),
                         Net([3, 5, 5], 300, n_blocks=2, n_filters=50).eval()),
        # 4 rows = 2 row-pair tiles per board: the 3-tile kernel must mask its third tile
        "bt4x5_2block": (games.load_game("breakthrough(rows=4,columns=5)"),
                         Net([3, 4, 5], 240, n_blocks=2, n_filters=40).eval()),
        # fc1 as a GEMM (az_head_gemm_kernel) with more filters than its 52-channel stride holds: the tower output keeps 64 channels
        "bt4x8_54f_2block": (games.load_game("breakthrough(rows=4,columns=8)"),
                             Net([3, 4, 8], 384, n_blocks=2, n_filters=54).eval()),
    }


@pytest.mark.parametrize("tag", ["c4_ckpt", "bt6_ckpt", "c4_10block", "bt5x4_3block", "bt4x5_2block"])
def test_packing_emulation_matches_torch(tag):
    game, net = _nets()[tag]
    boards = _random_boards(game, 5, 1)
    with torch.no_grad():
        p, v = net(torch.from_numpy(boards))
    packed = fusednet.pack_net(net)
    pe, ve, _ = fusednet.emulate_forward(packed, boards, round_fp16=True)
    assert np.abs(pe - p.numpy()).max() <= P_TOL
    assert np.abs(ve - v.numpy()[:, 0]).max() <= V_TOL


@pytest.mark.parametrize("tag", ["c4_ckpt", "bt6_ckpt", "bt4x5_2block"])
def test_split_packing_emulation_is_fp32_grade(tag):
    """The (hi, lo) fp16 pairs of the f16x3 path carry the weights to ~22 bits: the numpy emulation of the kernel's data
    movement with those pairs agrees with torch fp64 about as well as torch fp32 does."""
    game, net = _nets()[tag]
    boards = _random_boards(game, 5, 1)
    import copy
    with torch.no_grad():
        p64, v64 = copy.deepcopy(net).double()(torch.from_numpy(boards).double())
    packed = fusednet.pack_net(net)
    pe, ve, _ = fusednet.emulate_forward(packed, boards, split=True)
    assert np.abs(pe - p64.numpy()).max() <= 3e-6
    assert np.abs(ve - v64.numpy()[:, 0]).max() <= 3e-6
    x = np.array([1.0, 0.1, -3.3333333, 1234.567, 1e-3, 1e-5, 3e-8])
    hi, lo = fusednet.split_fp16(x)
    back = hi.astype(np.float64) + lo.astype(np.float64) / 2048.0
    assert np.abs(back[:5] / x[:5] - 1).max() < 2.0 ** -21     # ~22 significant bits in fp16's normal range
    assert np.abs(back[5:] - x[5:]).max() < 2e-11              # below it (fp16 subnormals) the ABSOLUTE error stays tiny


def test_net_matches_reference_golden_outputs():
    """Our Net class + the re-packed shipped checkpoints reproduce the reference Net.forward fixtures."""
    for tag, shape, A in [("connect_four", [3, 6, 7], 7), ("breakthrough6", [3, 6, 6], 432)]:
        net = load_npz_checkpoint(os.path.join(GOLDEN, "checkpoint_%s.npz" % tag), shape, A).eval()
        z = np.load(os.path.join(GOLDEN, "net_forward_%s.npz" % tag))
        with torch.no_grad():
            p, v = net(torch.from_numpy(z["boards"].astype(np.float32)))
        assert np.abs(p.numpy() - z["p"]).max() < 2e-5
        assert np.abs(v.numpy() - z["v"]).max() < 2e-5
        assert set(net.state_dict().keys()) == set(np.load(os.path.join(GOLDEN, "checkpoint_%s.npz" % tag)).files)


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["f32x", "f16"])
@pytest.mark.parametrize("tag,shape,A", [("connect_four", [3, 6, 7], 7), ("breakthrough6", [3, 6, 6], 432)])
def test_fused_forward_matches_the_reference_held_outputs(tag, shape, A, precision):
    """The HIP towers against the outputs the REFERENCE's own Net.forward (network.py:48-64) produced for its shipped checkpoints
    (tests/golden/net_forward_*.npz, written by oracle/gen_golden.py running the real reference): no torch module in between.
    f32x (the product default): 2e-5, the grade the fixture itself is pinned to on CPU; f16 (opt-in): P_TOL / V_TOL."""
    net = load_npz_checkpoint(os.path.join(GOLDEN, "checkpoint_%s.npz" % tag), shape, A).eval()
    z = np.load(os.path.join(GOLDEN, "net_forward_%s.npz" % tag))
    boards = z["boards"].astype(np.float32)
    n = boards.shape[0]
    fn = fusednet.FusedNet(net, "cuda:0", max_boards=max(n, 16), precision=precision)
    pf, vf = fn.forward(torch.from_numpy(boards).cuda())
    torch.cuda.synchronize()
    pf, vf = pf.cpu().numpy(), vf.cpu().numpy()
    fn.close()
    dp, dv = np.abs(pf - z["p"]).max(), np.abs(vf - z["v"].reshape(-1)).max()
    p_tol, v_tol = (2e-5, 2e-5) if precision == "f32x" else (P_TOL, V_TOL)
    assert dp <= p_tol and dv <= v_tol, (dp, dv)
    # the same boards inside a batch large enough for the full-batch kernels (x3b / az_tower_kernel): same bits as above
    big = np.concatenate([boards] * (600 // n + 1))[:600]
    fn = fusednet.FusedNet(net, "cuda:0", max_boards=600, precision=precision)
    pb, vb = fn.forward(torch.from_numpy(big).cuda())
    torch.cuda.synchronize()
    assert (pb.cpu().numpy()[:n] == pf).all() and (vb.cpu().numpy()[:n] == vf).all()
    fn.close()


@pytest.mark.gpu
@pytest.mark.parametrize("tag,n", [("c4_ckpt", 24), ("c4_ckpt", 1), ("c4_ckpt", 157), ("bt6_ckpt", 40),
                                   ("c4_10block", 300), ("bt8_2block", 37), ("bt5x4_3block", 50),
                                   ("bt4x5_2block", 3), ("bt4x5_2block", 130), ("c4_10block", 4096),
                                   ("c4_56f_2block", 33), ("c4_56f_2block", 2100), ("bt5x5_2block", 70),
                                   ("bt6_10block", 4096), ("bt8_20block", 2048),
                                   # the edges of az_head_gemm_kernel's 128-board tiles and 16-board MFMA tiles
                                   ("bt8_2block", 1), ("bt8_2block", 128), ("bt6_ckpt", 129), ("bt5x5_2block", 255),
                                   ("bt4x8_54f_2block", 140)])
def test_fused_forward_matches_torch(tag, n):
    game, net = _nets()[tag]
    boards = _random_boards(game, n, 7)
    with torch.no_grad():
        p, v = net(torch.from_numpy(boards))
    fn = fusednet.FusedNet(net, "cuda:0", max_boards=max(n, 16), precision="f16")
    obs = torch.from_numpy(boards).cuda()
    pf, vf = fn.forward(obs)
    torch.cuda.synchronize()
    pf, vf = pf.cpu().numpy(), vf.cpu().numpy()
    assert np.isfinite(pf).all() and np.isfinite(vf).all()
    assert np.abs(pf.sum(1) - 1).max() < 1e-5
    dp, dv = np.abs(pf - p.numpy()).max(), np.abs(vf - v.numpy()[:, 0]).max()
    assert dp <= P_TOL and dv <= V_TOL, (dp, dv)
    # the tower output (pre-fc residual stream) agrees with the fp16-rounding emulation of the same packing
    if n <= 40:
        _, _, tower = fusednet.emulate_forward(fn.packed, boards, round_fp16=True)
        got = fn.read_tower(n)
        scale = np.abs(tower).max() + 1e-6
        assert np.abs(got - tower).max() / scale < 5e-3
    # a second call with a different batch reuses the buffers
    pf2, vf2 = fn.forward(obs[: max(1, n // 2)].contiguous())
    torch.cuda.synchronize()
    assert np.abs(pf2.cpu().numpy() - pf[: max(1, n // 2)]).max() < 1e-6
    fn.close()


@pytest.mark.gpu
@pytest.mark.parametrize("tag,n", [("c4_ckpt", 157), ("c4_10block", 4096), ("bt6_ckpt", 40), ("bt6_10block", 4096),
                                   ("bt8_2block", 37), ("bt8_20block", 2048), ("bt5x4_3block", 50), ("bt4x5_2block", 130),
                                   ("c4_56f_2block", 33), ("bt5x5_2block", 70),
                                   ("bt8_2block", 1), ("bt8_2block", 128), ("bt6_ckpt", 129), ("bt5x5_2block", 255),
                                   ("bt4x8_54f_2block", 140)])
def test_fused_f32x_forward_is_fp32_grade(tag, n):
    """precision="f32x" (AZ_NET_PREC_F16X3: split-fp16 operands, three MFMAs per product) against an fp64 evaluation of
    the same net: the error must be of the order of torch-fp32's own error against fp64 - i.e. the path is a stand-in
    for the reference's fp32 Net.forward (network.py:48-64), not a reduced-precision approximation of it."""
    import copy
    game, net = _nets()[tag]
    boards = _random_boards(game, n, 7)
    with torch.no_grad():
        p32, v32 = net(torch.from_numpy(boards))
        p64, v64 = copy.deepcopy(net).double()(torch.from_numpy(boards).double())
    p64, v64 = p64.numpy(), v64.numpy()[:, 0]
    e32 = max(np.abs(p32.numpy() - p64).max(), np.abs(v32.numpy()[:, 0] - v64).max())
    fn = fusednet.FusedNet(net, "cuda:0", max_boards=max(n, 16), precision="f32x")
    pf, vf = fn.forward(torch.from_numpy(boards).cuda())
    torch.cuda.synchronize()
    pf, vf = pf.cpu().numpy().astype(np.float64), vf.cpu().numpy().astype(np.float64)
    assert np.isfinite(pf).all() and np.isfinite(vf).all() and np.abs(pf.sum(1) - 1).max() < 1e-5
    ex = max(np.abs(pf - p64).max(), np.abs(vf - v64).max())
    print("f32x %s n=%d: max err vs fp64 %.3g (torch fp32: %.3g)" % (tag, n, ex, e32))
    assert ex <= max(4.0 * e32, 2e-6), (ex, e32)
    if n <= 40:
        _, _, tower = fusednet.emulate_forward(fn.packed, boards, split=True)
        got = fn.read_tower(n)
        assert np.abs(got - tower).max() / (np.abs(tower).max() + 1e-6) < 2e-5
    fn.close()


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["c4_10block", "bt6_10block", "bt5x4_3block", "bt8_2block"])
def test_fused_f32x_outputs_do_not_depend_on_the_batch_size(tag):
    """The fp32-grade tower runs different kernels at different batch sizes: a board per workgroup for small batches
    (az_tower_x3c_kernel, <= 512 boards: four waves split a board by output-channel tile), a board per wave above
    (az_tower_x3b_kernel), and - 6x7 and 6x6 boards above 1024, 8x8 boards always - eight (four) boards packed into whole column
    tiles (az_tower_x3d_kernel).  Priors, value and tower output of a board must be the same BITS in all of them, whichever boards share
    its workgroup and however ragged the last one is: a generation's records may not depend on when its tail switches kernels."""
    game, net = _nets()[tag]
    packed = tag.startswith(("c4", "bt6", "bt8"))   # boards whose positions fill whole column tiles (az_tower_x3d.h)
    n_ref = 2048 if packed else 1024
    fn = fusednet.FusedNet(net, "cuda:0", max_boards=n_ref, precision="f32x")
    torch.manual_seed(5)
    obs = (torch.rand(n_ref, 4, game.rows, game.cols, device="cuda") > 0.5).float()
    ref_p, ref_v = [t.clone() for t in fn.forward(obs)]
    torch.cuda.synchronize()
    ref_t = fn.read_tower(n_ref)
    assert ("x3d" if packed else "x3b") in fn.kernel_label(n_ref)
    if tag.startswith(("c4", "bt6")):
        assert "x3d" in fn.kernel_label(1500) and "x3b" in fn.kernel_label(1024) and "x3c" in fn.kernel_label(300)
    elif not tag.startswith("bt8"):
        assert "x3c" in fn.kernel_label(300)
    for n in (1500, 1024, 700, 512, 300, 256, 64, 5, 1):
        if n > n_ref:
            continue
        p, v = fn.forward(obs[:n].contiguous())
        torch.cuda.synchronize()
        assert torch.equal(p, ref_p[:n]) and torch.equal(v, ref_v[:n]), n
        assert (fn.read_tower(n) == ref_t[:n]).all(), n
    fn.close()


@pytest.mark.gpu
def test_fused_outputs_do_not_depend_on_the_batch_size():
    """Different batch sizes run different kernel variants (8-wave / 4-wave workgroups, chunk sizes, 1-2 boards per wave):
    a board's priors, value and tower output must be the same bits in all of them (the engine's slot-count invariance
    rests on this)."""
    game, net = _nets()["c4_10block"]
    fn = fusednet.FusedNet(net, "cuda:0", max_boards=4096, precision="f16")
    torch.manual_seed(5)
    obs = (torch.rand(4096, 4, 6, 7, device="cuda") > 0.5).float()
    ref_p, ref_v = [t.clone() for t in fn.forward(obs)]
    torch.cuda.synchronize()
    ref_t = fn.read_tower(4096)
    for n in (2048, 1024, 513, 512, 300, 257, 256, 64, 5):   # (<= 512 / <= 256 boards: az_tower_f16c_kernel, 16- / 32-KiB weight chunks)
        p, v = fn.forward(obs[:n].contiguous())
        torch.cuda.synchronize()
        assert torch.equal(p, ref_p[:n]) and torch.equal(v, ref_v[:n]), n
        assert (fn.read_tower(n) == ref_t[:n]).all(), n
    fn.close()


@pytest.mark.gpu
def test_fused_net_rejects_bad_arguments():
    game, net = _nets()["c4_ckpt"]
    fn = fusednet.FusedNet(net, "cuda:0", max_boards=8, precision="f16")
    with pytest.raises(RuntimeError):
        fn.forward(torch.zeros(9, 4, 6, 7, device="cuda"))      # more boards than reserved
    with pytest.raises(RuntimeError):
        fn.forward(torch.zeros(4, 4, 6, 7, device="cuda", dtype=torch.float16))
    with pytest.raises(ValueError):
        fusednet.pack_net(Net([3, 6, 7], 7, n_blocks=2, n_filters=64).eval())  # > 56 filters
    fn.close()
