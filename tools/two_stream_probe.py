#!/usr/bin/env python3
"""Experiment: R engine replicas of G/R slots each on R HIP streams of ONE GPU (independent games, no sync between
them) versus one engine of G slots.  Measures whole-GPU games/s on the headline workload."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphazero_openspiel_amd import engine as E, fusednet  # noqa: E402
from alphazero_openspiel_amd.network import Net  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--replicas", type=int, default=2)
ap.add_argument("--slots", type=int, default=4096, help="total over replicas")
ap.add_argument("--games", type=int, default=8192, help="total over replicas")
ap.add_argument("--playouts", type=int, default=400)
ap.add_argument("--check-every", type=int, default=32)
a = ap.parse_args()
R = a.replicas
torch.manual_seed(0)
net = Net([3, 6, 7], 7, n_blocks=10, n_filters=50).eval()
dev = torch.device("cuda:0")
reps = []
for r in range(R):
    st = torch.cuda.Stream(dev)
    with torch.cuda.stream(st):
        fn = fusednet.FusedNet(net, dev, max_boards=a.slots // R, precision="f16")
        eng = E.SelfPlayEngine("connect_four", a.slots // R, n_playouts=a.playouts, max_games=a.games // R, seed=100 + r, device=0)
        eng.reset(a.games // R)
        obs, pri, val = eng.alloc_io()
        for _ in range(2):
            eng.advance(pri, val, obs)
            fn(obs, pri, val)
    st.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        eng.advance(pri, val, obs)
        fn(obs, pri, val)
    reps.append((st, fn, eng, g, (obs, pri, val)))
torch.cuda.synchronize()
t0 = time.perf_counter()
done = [False] * R
ticks = 0
while not all(done):
    for _ in range(a.check_every):
        for r, (st, fn, eng, g, io) in enumerate(reps):
            if not done[r]:
                with torch.cuda.stream(st):
                    g.replay()
        ticks += 1
    for r, (st, fn, eng, g, io) in enumerate(reps):
        if not done[r]:
            with torch.cuda.stream(st):
                done[r] = eng.games_done() >= a.games // R
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("replicas=%d slots=%d games=%d: %.3f s -> %.0f games/s, %d ticks, %.1f us/tick-round" %
      (R, a.slots, a.games, dt, a.games / dt, ticks, 1e6 * dt / ticks))
