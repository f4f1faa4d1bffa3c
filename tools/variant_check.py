#!/usr/bin/env python3
"""Debug aid: the fused net must give bit-identical outputs for a board whatever the batch size (kernel variant)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphazero_openspiel_amd.fusednet import FusedNet  # noqa: E402
from alphazero_openspiel_amd.network import Net  # noqa: E402

torch.manual_seed(0)
net = Net([3, 6, 7], 7, n_blocks=10, n_filters=50).eval()
fn = FusedNet(net, "cuda:0", max_boards=4096, precision="f16")
obs = (torch.rand(4096, 4, 6, 7, device="cuda") > 0.5).float()
ref_p, ref_v = [t.clone() for t in fn.forward(obs)]
torch.cuda.synchronize()
ref_t = fn.read_tower(4096)
for n in (4096, 2048, 1024, 300, 64, 5):
    p, v = fn.forward(obs[:n].contiguous())
    torch.cuda.synchronize()
    t = fn.read_tower(n)
    print(n, "dprior", float((p - ref_p[:n]).abs().max()), "dvalue", float((v - ref_v[:n]).abs().max()),
          "dtower", float(np.abs(t - ref_t[:n]).max()), "rows differing", int((np.abs(t - ref_t[:n]).reshape(n, -1).max(1) > 0).sum()))
