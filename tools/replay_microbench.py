#!/usr/bin/env python3
"""Times the device replay path: engine -> append -> dedupe -> 500 x (sample + net_step), i.e. one
Trainer.train_network (train.py:132-154) worth of work, and the same with the restated host path for a small slice."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphazero_openspiel_amd import engine as E, games, replay  # noqa: E402
from alphazero_openspiel_amd.fusednet import FusedNet  # noqa: E402
from alphazero_openspiel_amd.network import Net  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--games", type=int, default=4096)
ap.add_argument("--playouts", type=int, default=50)
ap.add_argument("--batches", type=int, default=500)
a = ap.parse_args()
g = games.load_game("connect_four")
torch.manual_seed(0)
net = Net([3, 6, 7], 7).cuda()
eng = E.SelfPlayEngine(g, min(a.games, 4096), n_playouts=a.playouts, max_games=a.games, seed=1)
t = time.perf_counter(); E.run_selfplay(eng, FusedNet(net.eval(), "cuda:0", max_boards=eng.G, precision="f16"), a.games, use_graph=True); torch.cuda.synchronize()
t_play = time.perf_counter() - t
rep = replay.DeviceReplay(g, max_games=a.games)
t = time.perf_counter(); rep.append_engine(eng); torch.cuda.synchronize(); t_app = time.perf_counter() - t
t = time.perf_counter(); n_u = rep.dedupe(); torch.cuda.synchronize(); t_ded = time.perf_counter() - t
st = rep.stats()
opt = replay.make_optimizer(net)
net.train()
t = time.perf_counter()
for i in range(a.batches):
    x, pi, z = rep.sample(256, seed=7)
    replay.net_step(net, opt, x, pi, z)
torch.cuda.synchronize()
t_train = time.perf_counter() - t
bytes_ex = 8 + 16 + 4 + 8 + 8 * 7
print("self-play %d games: %.2f s | append %d examples: %.1f ms | dedupe -> %d unique: %.1f ms (%.1f M examples/s, %.2f GB/s of records) | "
      "%d x (gather 256 + net_step): %.2f s (%.2f ms/step)" % (a.games, t_play, st["n_examples"], 1e3 * t_app, n_u, 1e3 * t_ded,
                                                            st["n_examples"] / t_ded / 1e6, st["n_examples"] * bytes_ex / t_ded / 1e9,
                                                            a.batches, t_train, 1e3 * t_train / a.batches))
gs = replay.GraphedNetStep(net, 256, rep)
torch.cuda.synchronize()
t = time.perf_counter()
for i in range(a.batches):
    gs(seed=9)
torch.cuda.synchronize()
t_g = time.perf_counter() - t
print("graphed: %d x (gather 256 + net_step): %.2f s (%.2f ms/step), last losses %.4f %.4f" % (a.batches, t_g, 1e3 * t_g / a.batches, float(gs.loss_p), float(gs.loss_v)))
# host path for comparison: export -> python lists -> restated remove_duplicates (what train.py does)
t = time.perf_counter(); ex = eng.export(); host = E.examples_from_export(g, ex); t_list = time.perf_counter() - t
print("host path for comparison: export + reference-format python lists %.2f s (the reference's remove_duplicates then walks them in "
      "Python; the CPU restatement of it is test infrastructure - tests/test_replay.py times nothing)" % t_list)
