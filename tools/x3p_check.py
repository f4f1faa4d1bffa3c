"""az_tower_x3p_kernel (two waves per board) against az_tower_x3b_kernel (AZ_NET_TOWER=x3b) and az_tower_x3c_kernel (<= 512 boards):
the same BITS, and the time of each on the same box.  python tools/x3p_check.py [--blocks 10] [--iters 30]"""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from alphazero_openspiel_amd import games
from alphazero_openspiel_amd.fusednet import FusedNet
from alphazero_openspiel_amd.network import Net

ap = argparse.ArgumentParser()
ap.add_argument("--blocks", type=int, default=10)
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--boards", type=int, default=4096)
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


for game, blocks in (("connect_four", a.blocks), ("breakthrough(rows=6,columns=6)", 2), ("breakthrough(rows=5,columns=4)", 3)):
    g = games.load_game(game)
    torch.manual_seed(0)
    net = Net(g.information_state_normalized_vector_shape(), g.num_distinct_actions(), n_blocks=blocks, n_filters=50).eval()
    B = a.boards
    obs = (torch.rand(B, 4, g.rows, g.cols, device="cuda") > 0.5).float()
    pri = torch.empty(B, g.num_distinct_actions(), device="cuda")
    val = torch.empty(B, device="cuda")
    out = {}
    for tag in ("x3p", "x3b"):
        if tag == "x3b":
            os.environ["AZ_NET_TOWER"] = "x3b"
        else:
            os.environ.pop("AZ_NET_TOWER", None)
        fn = FusedNet(net, "cuda:0", max_boards=B, precision="f32x")
        p, v = fn.forward(obs)
        torch.cuda.synchronize()
        out[tag] = (p.cpu().numpy().copy(), v.cpu().numpy().copy(), fn.read_tower(B).copy())
        med, mn = timed(fn, obs, pri, val, a.iters)
        print("%s %d blocks, %d boards, %s: tower + head median %.1f us, min %.1f us" % (game, blocks, B, tag, med, mn), flush=True)
        if tag == "x3p":
            for n in (B - 3, 1500, 700):  # ragged last workgroups
                ps, vs = fn.forward(obs[:n].contiguous())
                torch.cuda.synchronize()
                ok = bool((ps.cpu().numpy() == out[tag][0][:n]).all() and (vs.cpu().numpy() == out[tag][1][:n]).all())
                print("   %d boards vs the first %d of %d: %s" % (n, n, B, "bit-identical" if ok else "DIFFERENT"))
            ps, vs = fn.forward(obs[:300].contiguous())  # x3c
            torch.cuda.synchronize()
            ok = bool((ps.cpu().numpy() == out[tag][0][:300]).all() and (vs.cpu().numpy() == out[tag][1][:300]).all())
            print("   300 boards (az_tower_x3c_kernel): %s" % ("bit-identical" if ok else "DIFFERENT"))
        fn.close()
    os.environ.pop("AZ_NET_TOWER", None)
    same = all((out["x3p"][i] == out["x3b"][i]).all() for i in range(3))
    print("%s: x3p vs x3b %s" % (game, "bit-identical" if same else "DIFFERENT max|d tower| %.3g, max|d prior| %.3g"
                                 % (np.abs(out["x3p"][2] - out["x3b"][2]).max(), np.abs(out["x3p"][0] - out["x3b"][0]).max())), flush=True)
