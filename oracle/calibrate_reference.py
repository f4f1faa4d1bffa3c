"""TEST / BASELINE INFRASTRUCTURE (build container only: needs /root/reference) — BASELINE.md section 3 step 1.

Times the REAL reference's in-process self-play (`game_utils.play_game_self` with `policy_fn = Net.predict`, one CPU thread,
the shipped connect_four checkpoint) and the C restatement with the same network on the same settings, and prints the
calibration ratio restatement / reference.  With it, a CPU-restatement figure measured on the GPU box translates into an
implied figure for the Python reference there.

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.calibrate_reference [--games25 6] [--games400 2]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import binding as orc, pygames, ref_harness  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games25", type=int, default=6)
    ap.add_argument("--games400", type=int, default=2)
    a = ap.parse_args()
    torch.set_num_threads(1)
    ref = ref_harness.load_reference()
    game_name = "connect_four"
    game = pygames.load_game(game_name)
    shape, A = game.information_state_normalized_vector_shape(), game.num_distinct_actions()
    net = ref.network.Net(shape, A)
    net.load_state_dict(torch.load(os.path.join(ref_harness.REFERENCE_DIR, "models", "example_model_connect_four.pth"),
                                   map_location="cpu", weights_only=True))
    net.eval()

    def policy(board):  # the restatement's evaluator: the same module, the same batch-1 forward
        with torch.no_grad():
            p, v = net(torch.from_numpy(board.reshape(1, 4, 6, 7)).float())
        return p[0].double().numpy(), float(v)

    out = {}
    for S, n in ((25, a.games25), (400, a.games400)):
        np.random.seed(1)
        t = time.perf_counter()
        plies_ref = 0
        for _ in range(n):
            plies_ref += len(ref.game_utils.play_game_self(net.predict, game_name, n_playouts=S, temperature=1.0, dirichlet_ratio=0.25,
                                                           c_puct=2.5, backup="on-policy"))
        t_ref = time.perf_counter() - t
        t = time.perf_counter()
        plies_orc, sims = 0, 0
        for k in range(n):
            o = orc.play_game_self(policy, game_name, n_playouts=S, seed=100 + k)
            plies_orc += len(o["actions"])
            sims += o["counters"]["sims"]
        t_orc = time.perf_counter() - t
        out["S=%d" % S] = {"games": n, "reference_plies_per_s": plies_ref / t_ref, "restatement_plies_per_s": plies_orc / t_orc,
                           "reference_games_per_s": n / t_ref, "restatement_games_per_s": n / t_orc,
                           "ratio_restatement_over_reference_per_ply": (plies_orc / t_orc) / (plies_ref / t_ref)}
    out["where"] = "build container, 1 thread, %s" % open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
