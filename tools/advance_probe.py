#!/usr/bin/env python3
"""How long does az_engine_advance take when NO slot is in its move step (fresh engine: all slots in lock-step, the first
moves come after ~S ticks) and when slots are moving (desynchronised steady state)?  HIP events around the launch."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphazero_openspiel_amd import engine as E  # noqa: E402
from alphazero_openspiel_amd.fusednet import FusedNet  # noqa: E402
from alphazero_openspiel_amd.network import Net  # noqa: E402

game = sys.argv[1] if len(sys.argv) > 1 else "connect_four"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 400
G = 4096
kw = {}
for a in sys.argv[3:]:  # extra engine keywords, e.g. use_dirichlet=0 backup=soft-Z
    k, v = a.split("=")
    kw[k] = int(v) if v.lstrip("-").isdigit() else v
print("engine keywords:", kw)
eng = E.SelfPlayEngine(game, G, n_playouts=S, max_games=8 * G, device=0, seed=1, **kw)
torch.manual_seed(1)
net = Net(eng.game.information_state_normalized_vector_shape(), eng.A, n_blocks=10, n_filters=50).eval()
fn = FusedNet(net, "cuda:0", max_boards=G, precision="f16")
eng.reset(8 * G)
obs, pri, val = eng.alloc_io()


def timed_ticks(n):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    m0 = eng.progress()["moves"]
    for s, e in ev:
        s.record()
        eng.advance(pri, val, obs)
        e.record()
        fn(obs, pri, val)
    torch.cuda.synchronize()
    t = np.array([s.elapsed_time(e) for s, e in ev]) * 1e3
    return t, eng.progress()["moves"] - m0


for _ in range(20):
    eng.advance(pri, val, obs)
    fn(obs, pri, val)
t, mv = timed_ticks(200)
print("lock-step phase, ticks 20-220 (moves in window: %d): advance median %.1f us  p90 %.1f  max %.1f" % (mv, np.median(t), np.quantile(t, 0.9), t.max()))
for _ in range(40):  # run a few thousand ticks to desynchronise
    for _ in range(100):
        eng.advance(pri, val, obs)
        fn(obs, pri, val)
    torch.cuda.synchronize()
t, mv = timed_ticks(200)
print("steady state (moves in window: %d = %.1f per tick): advance median %.1f us  p90 %.1f  max %.1f" % (mv, mv / 200.0, np.median(t), np.quantile(t, 0.9), t.max()))
