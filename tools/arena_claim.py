import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphazero_openspiel_amd import arena
from alphazero_openspiel_amd.network import load_npz_checkpoint
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
net = load_npz_checkpoint(os.path.join(ROOT, "tests/golden/checkpoint_breakthrough6.npz"), [3, 6, 6], 432)
t = time.time()
s1, s2, prog = arena.play_tests(net, "breakthrough(rows=6,columns=6)", 512, "zero", "uct", opponent_sims=200, device="cuda:0", seed=11, n_playouts=100, c_puct=2.5)
wins = int((s1 > 0).sum() + (s2 > 0).sum())
print("bt6 AlphaZero@100 vs UCT@200: %d / 1024 games won (%.2f %%), avg reward %.4f, %.1f s" % (wins, 100.0 * wins / 1024, (s1.sum() + s2.sum()) / 1024, time.time() - t))
net4 = load_npz_checkpoint(os.path.join(ROOT, "tests/golden/checkpoint_connect_four.npz"), [3, 6, 7], 7)
s1, s2, prog = arena.play_tests(net4, "connect_four", 512, "zero", "uct", opponent_sims=200, device="cuda:0", seed=12, n_playouts=100)
wins = int((s1 > 0).sum() + (s2 > 0).sum()); draws = int((s1 == 0).sum() + (s2 == 0).sum())
print("c4 AlphaZero@100 vs UCT@200: %d won, %d drawn of 1024, avg reward %.4f" % (wins, draws, (s1.sum() + s2.sum()) / 1024))
