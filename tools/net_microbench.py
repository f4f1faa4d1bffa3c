#!/usr/bin/env python3
"""Times az_net_forward alone (HIP events on the launch stream) — the tower/head kernels in isolation.
    python tools/net_microbench.py [--boards 4096] [--blocks 10] [--iters 50] [--game connect_four]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphazero_openspiel_amd import _lib, games  # noqa: E402
from alphazero_openspiel_amd.fusednet import FusedNet  # noqa: E402
from alphazero_openspiel_amd.network import Net  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--boards", type=int, default=4096)
ap.add_argument("--blocks", type=int, default=10)
ap.add_argument("--filters", type=int, default=50)
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--game", default="connect_four")
ap.add_argument("--precision", default="f16", choices=["f16", "f32x"])
ap.add_argument("--zero", action="store_true", help="zero weights and zero boards: the same instruction stream on all-zero MFMA operands "
                                                    "(DVFS check: does the chip run the identical kernel faster when it draws less power?)")
ap.add_argument("--lib", default=None, help="an experimental build of the engine library (tools only; the product loads its own)")
a = ap.parse_args()
if a.lib:
    _lib.LIB_PATH = os.path.abspath(a.lib)
g = games.load_game(a.game)
torch.manual_seed(0)
net = Net(g.information_state_normalized_vector_shape(), g.num_distinct_actions(), n_blocks=a.blocks, n_filters=a.filters).eval()
if a.zero:
    with torch.no_grad():
        for prm in net.parameters():
            prm.zero_()
fn = FusedNet(net, "cuda:0", max_boards=a.boards, precision=a.precision)
obs = (torch.rand(a.boards, 4, g.rows, g.cols, device="cuda") > 0.5).float()  # random 0/1 planes
if a.zero:
    obs.zero_()
pri = torch.empty(a.boards, g.num_distinct_actions(), device="cuda")
val = torch.empty(a.boards, device="cuda")
for _ in range(5):
    fn(obs, pri, val)
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.iters)]
for s, e in ev:
    s.record()
    fn(obs, pri, val)
    e.record()
torch.cuda.synchronize()
t = np.array([s.elapsed_time(e) for s, e in ev])
H, W, A = g.rows, g.cols, g.num_distinct_actions()
F0 = a.filters
flops = 2 * H * W * (9 * 4 * F0 + 4 * F0 + (2 * a.blocks - 1) * 9 * F0 * F0) + 2 * F0 * H * W * (A + 1)
print("boards=%d blocks=%d: median %.1f us  min %.1f us  -> %.1f TFLOP/s algorithmic" %
      (a.boards, a.blocks, 1e3 * np.median(t), 1e3 * t.min(), a.boards * flops / (np.median(t) * 1e-3) / 1e12))
