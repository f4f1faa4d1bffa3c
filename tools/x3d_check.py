"""az_tower_x3d_kernel (packed column tiles) against the kernels of round 3 (AZ_NET_TOWER=x3b: az_tower_x3b_kernel on row-pair boards -
the same BITS are expected - and az_tower_x3_kernel on 8x8 - fp32-grade agreement) and az_tower_x3c_kernel (<= 512 boards), with the
time of each on the same box.  python tools/x3d_check.py [--iters 30] [--only c4|bt6|bt8]"""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from alphazero_openspiel_amd import games
from alphazero_openspiel_amd.fusednet import FusedNet
from alphazero_openspiel_amd.network import Net

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--only", default=None)
a = ap.parse_args()


def timed(fn, obs, pri, val, iters):
    for _ in range(5):
        fn(obs, pri, val)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in ev:
        s.record()
        fn(obs, pri, val)
        e.record()
    torch.cuda.synchronize()
    t = np.array([s.elapsed_time(e) for s, e in ev])
    return 1e3 * np.median(t), 1e3 * t.min()


CASES = {"c4": ("connect_four", 10, 4096), "bt6": ("breakthrough(rows=6,columns=6)", 10, 4096),
         "bt8": ("breakthrough(rows=8,columns=8)", 20, 2048), "c4_3": ("connect_four", 3, 4096),
         "bt6_1": ("breakthrough(rows=6,columns=6)", 1, 64), "bt6_2": ("breakthrough(rows=6,columns=6)", 2, 640)}
for tag, (game, blocks, B) in CASES.items():
    if a.only and tag != a.only:
        continue
    g = games.load_game(game)
    torch.manual_seed(0)
    net = Net(g.information_state_normalized_vector_shape(), g.num_distinct_actions(), n_blocks=blocks, n_filters=50).eval()
    obs = (torch.rand(B, 4, g.rows, g.cols, device="cuda") > 0.5).float()
    pri = torch.empty(B, g.num_distinct_actions(), device="cuda")
    val = torch.empty(B, device="cuda")
    out = {}
    for kern in ("x3d", "x3b"):
        if kern == "x3b":
            os.environ["AZ_NET_TOWER"] = "x3b"
        else:
            os.environ.pop("AZ_NET_TOWER", None)
        fn = FusedNet(net, "cuda:0", max_boards=B, precision="f32x")
        p, v = fn.forward(obs)
        torch.cuda.synchronize()
        out[kern] = (p.cpu().numpy().copy(), v.cpu().numpy().copy(), fn.read_tower(B).copy())
        med, mn = timed(fn, obs, pri, val, a.iters)
        print("%s %d blocks, %d boards, %s: tower + head median %.1f us, min %.1f us" % (game, blocks, B, kern, med, mn), flush=True)
        if kern == "x3d":
            for n in [m for m in (B - 3, 1501, 700, 300, 5) if m < B]:  # ragged last workgroups; <= 512 on row-pair boards: az_tower_x3c_kernel
                ps, vs = fn.forward(obs[:n].contiguous())
                torch.cuda.synchronize()
                ok = bool((ps.cpu().numpy() == out[kern][0][:n]).all() and (vs.cpu().numpy() == out[kern][1][:n]).all())
                print("   %d boards vs the first %d of %d: %s" % (n, n, B, "bit-identical" if ok else "DIFFERENT max|d| %.3g"
                                                               % np.abs(ps.cpu().numpy() - out[kern][0][:n]).max()), flush=True)
        fn.close()
    os.environ.pop("AZ_NET_TOWER", None)
    same = all((out["x3d"][i] == out["x3b"][i]).all() for i in range(3))
    print("%s: x3d vs round-3 kernel: %s" % (game, "bit-identical" if same else "max|d tower| %.3g (rel %.3g), max|d prior| %.3g, max|d value| %.3g"
          % (np.abs(out["x3d"][2] - out["x3b"][2]).max(), np.abs(out["x3d"][2] - out["x3b"][2]).max() / np.abs(out["x3b"][2]).max(),
             np.abs(out["x3d"][0] - out["x3b"][0]).max(), np.abs(out["x3d"][1] - out["x3b"][1]).max())), flush=True)
