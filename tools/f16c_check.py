import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from alphazero_openspiel_amd import games
from alphazero_openspiel_amd.fusednet import FusedNet
from alphazero_openspiel_amd.network import Net
for game, blocks, filters in (("connect_four", 3, 50), ("breakthrough(rows=6,columns=6)", 2, 50), ("breakthrough(rows=5,columns=4)", 2, 32), ("connect_four", 2, 56)):
    g = games.load_game(game)
    torch.manual_seed(0)
    net = Net(g.information_state_normalized_vector_shape(), g.num_distinct_actions(), n_blocks=blocks, n_filters=filters).eval()
    fn = FusedNet(net, "cuda:0", max_boards=2048, precision="f16")
    obs = (torch.rand(2048, 4, g.rows, g.cols, device="cuda") > 0.5).float()
    pb, vb = fn.forward(obs)            # 2048 boards: az_tower_kernel
    tb = fn.read_tower(2048).copy()
    for n in (1, 7, 256, 512):
        ps, vs = fn.forward(obs[:n].contiguous())   # <= 512 boards: az_tower_f16c_kernel where the board qualifies
        ts = fn.read_tower(n)
        same = bool((ps == pb[:n]).all() and (vs == vb[:n]).all() and (ts == tb[:n]).all())
        print(game, blocks, filters, n, "bit-identical" if same else "DIFFERENT max|d| %.3g" % np.abs(ts - tb[:n]).max(), flush=True)
