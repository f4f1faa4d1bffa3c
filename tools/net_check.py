#!/usr/bin/env python3
"""Quick numerics check of a (possibly experimental) build of the engine library against the torch module:
    python tools/net_check.py [--lib path/to/libaz_engine.so] [--precision f32x] [--blocks 3] [--boards 300]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--precision", default="f32x")
ap.add_argument("--blocks", type=int, default=3)
ap.add_argument("--boards", type=int, default=300)
ap.add_argument("--game", default="connect_four")
ap.add_argument("--save", default=None, help="np.save the tower output (pre-fc residual stream) here")
ap.add_argument("--ref", default=None, help="compare the tower output with this saved one, per channel")
a = ap.parse_args()
from alphazero_openspiel_amd import _lib  # noqa: E402
if a.lib:
    _lib.LIB_PATH = os.path.abspath(a.lib)
from alphazero_openspiel_amd import games  # noqa: E402
from alphazero_openspiel_amd.fusednet import FusedNet  # noqa: E402
from alphazero_openspiel_amd.network import Net  # noqa: E402

g = games.load_game(a.game)
torch.manual_seed(0)
net = Net(g.information_state_normalized_vector_shape(), g.num_distinct_actions(), n_blocks=a.blocks, n_filters=50).eval()
with torch.no_grad():  # non-trivial BatchNorm statistics
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.2)
            m.running_var.uniform_(0.5, 1.5)
obs = (torch.rand(a.boards, 4, g.rows, g.cols) > 0.5).float()
with torch.no_grad():
    p, v = net(obs)
fn = FusedNet(net, "cuda:0", max_boards=a.boards, precision=a.precision)
pf, vf = fn.forward(obs.cuda())
torch.cuda.synchronize()
dp = (pf.cpu() - p).abs().max().item()
dv = (vf.cpu() - v[:, 0]).abs().max().item()
print("%s %s blocks=%d boards=%d: max|dprior| %.3g  max|dvalue| %.3g  finite=%s" %
      (a.lib or "default", a.precision, a.blocks, a.boards, dp, dv, bool(torch.isfinite(pf).all() and torch.isfinite(vf).all())))
if a.save or a.ref:
    tw = fn.read_tower(a.boards)  # [boards][HW][64]
    if a.save:
        np.save(a.save, tw)
    if a.ref:
        ref = np.load(a.ref)
        d = np.abs(tw - ref)
        print("tower max|d| %.3g; per channel max: %s" % (d.max(), np.array2string(d.max(axis=(0, 1))[:50], precision=1, max_line_width=250)))
        print("per position max: %s" % np.array2string(d.max(axis=(0, 2)), precision=1, max_line_width=250))
