"""TEST / BASELINE INFRASTRUCTURE — full self-play games on host cores with the C oracle.

What the reference does in one worker process (examplegenerator.py:17-22 -> game_utils.py:148-206 ->
mcts.py:126-190): strictly sequential playouts, one batch-1 `Net.forward` per leaf on the CPU
(torch, 1 thread).  `python -m oracle.cpu_selfplay --games N ...` plays N FULL games and prints one JSON
line; bench.py's cpu_baseline leg starts one such process per host core (never imported by the product).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def play(game_name, n_games, S, n_blocks, n_filters, seed, handshake=False, budget_s=None, weight_seed=1, checkpoint=None):
    import numpy as np
    import torch

    from alphazero_openspiel_amd.games import Game
    from alphazero_openspiel_amd.network import Net
    from oracle import binding as orc

    torch.set_num_threads(1)
    g = Game(game_name)
    A = g.num_distinct_actions()
    torch.manual_seed(weight_seed)  # same weights in every worker, and as bench.py's GPU net (torch.manual_seed(args.seed))
    if checkpoint:  # the reference's shipped weights (tests/golden/checkpoint_*.npz): realistic game lengths and tree shapes
        from alphazero_openspiel_amd.network import load_npz_checkpoint
        net = load_npz_checkpoint(checkpoint, g.information_state_normalized_vector_shape(), A).eval()
    else:
        net = Net(g.information_state_normalized_vector_shape(), A, n_blocks=n_blocks, n_filters=n_filters).eval()

    def policy(board):
        with torch.no_grad():
            p, v = net(torch.from_numpy(board.reshape(1, 4, g.rows, g.cols)).float())
        return p[0].double().numpy(), float(v)

    policy(np.zeros(4 * g.rows * g.cols))  # warm-up (allocator, oneDNN primitive cache)
    orc.lib()
    if handshake:  # all workers of one leg start together, after every import and warm-up is done
        print("READY", flush=True)
        sys.stdin.readline()
    tot = {"sims": 0, "evals": 0, "plies": 0}
    t0, t_wall0 = time.perf_counter(), time.time()
    played = 0
    for k in range(n_games):
        if budget_s is not None and played and time.perf_counter() - t0 > budget_s:
            break  # bounded sample: only whole games are counted
        played += 1
        out = orc.play_game_self(policy, game_name, n_playouts=S, seed=seed * 1000003 + k)
        tot["sims"] += out["counters"]["sims"]
        tot["evals"] += out["counters"]["evals"]
        tot["plies"] += len(out["actions"])
    dt = time.perf_counter() - t0
    return {"games": played, "seconds": dt, "t_start": t_wall0, "t_end": time.time(), **tot}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--game", default="connect_four")
    ap.add_argument("--games", type=int, default=1)
    ap.add_argument("--playouts", type=int, default=400)
    ap.add_argument("--blocks", type=int, default=10)
    ap.add_argument("--filters", type=int, default=50)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--weight-seed", type=int, default=1)
    ap.add_argument("--budget-s", type=float, default=None, help="stop starting new games after this many seconds")
    ap.add_argument("--checkpoint", default=None, help="npz checkpoint to load instead of random-initialised weights")
    ap.add_argument("--handshake", action="store_true", help="print READY, then wait for a line on stdin before playing")
    a = ap.parse_args()
    print(json.dumps(play(a.game, a.games, a.playouts, a.blocks, a.filters, a.seed, a.handshake, a.budget_s, a.weight_seed, a.checkpoint)))


if __name__ == "__main__":
    main()
