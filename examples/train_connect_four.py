#!/usr/bin/env python3
"""The reference's Trainer.run loop (train.py:272-293) with every stage on the GPU:

    generation:  self-play (HIP search + fused net)  ->  device replay store (FIFO, remove_duplicates)
                 ->  n_batches x (gather 256 + net_step, replayed as one HIP graph)
                 ->  every --eval-every generations: Trainer.test_agent (train.py:238-270) on the device arena

    python examples/train_connect_four.py --generations 3 --games 512 --playouts 100

Hyper-parameters default to the reference's (train.py:24-49): 100 playouts/move, c_puct 2.5, temperature 1,
Dirichlet ratio 0.25, 500 batches of 256 per generation, Adam lr 1e-3 wd 1e-4, buffer of 4 generations growing to 40.
Checkpoints are written in the reference's .pth format.
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphazero_openspiel_amd import arena, engine as E, games, replay  # noqa: E402
from alphazero_openspiel_amd.fusednet import FusedNet  # noqa: E402
from alphazero_openspiel_amd.network import Net  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--game", default="connect_four")
    ap.add_argument("--generations", type=int, default=3)
    ap.add_argument("--games", type=int, default=500, help="n_games_per_generation (train.py:36)")
    ap.add_argument("--playouts", type=int, default=100, help="n_playouts_train (train.py:46)")
    ap.add_argument("--batches", type=int, default=500, help="n_batches_per_generation (train.py:37)")
    ap.add_argument("--batch-size", type=int, default=256)
    ap.add_argument("--backup", default="on-policy", choices=["on-policy", "soft-Z", "A0C", "off-policy"])
    ap.add_argument("--save", default=None, help="directory for <generation>.pth checkpoints")
    ap.add_argument("--precision", default="f32x", choices=["f32x", "f16"], help="fused-net arithmetic (f32x = fp32-grade)")
    ap.add_argument("--eval-every", type=int, default=0, help="run test_agent every N generations (0 = never)")
    ap.add_argument("--tests", type=int, default=200, help="n_tests (train.py:30)")
    a = ap.parse_args()

    dev = torch.device("cuda:0")
    game = games.load_game(a.game)
    torch.manual_seed(0)
    net = Net(game.information_state_normalized_vector_shape(), game.num_distinct_actions()).to(dev)
    n_buffer, n_buffer_max = 4 * a.games, 40 * a.games                     # train.py:38-41
    store = replay.DeviceReplay(game, max_games=n_buffer_max, device=dev)
    trainer = None
    for gen in range(1, a.generations + 1):
        t0 = time.perf_counter()
        net.eval()
        eng = E.SelfPlayEngine(game, min(a.games, 4096), n_playouts=a.playouts, backup=a.backup, max_games=a.games,
                               device=dev, seed=gen)
        prog = E.run_selfplay(eng, FusedNet(net, dev, max_boards=eng.G, precision=a.precision), a.games, use_graph=True)
        t_play = time.perf_counter() - t0
        if gen % 2 == 0 and n_buffer < n_buffer_max:                       # Trainer.update_buffer_size
            n_buffer += a.games
        store.append_engine(eng)
        eng.close()
        store.set_capacity(n_buffer)
        n_unique = store.dedupe()
        net.train()
        if trainer is None:
            trainer = replay.GraphedNetStep(net, a.batch_size, store)
        t1 = time.perf_counter()
        lp = lv = 0.0
        for i in range(a.batches):
            p, v = trainer(seed=gen)
            if i >= a.batches - 100:
                lp += float(p) / 100
                lv += float(v) / 100
        torch.cuda.synchronize()
        t_train = time.perf_counter() - t1
        st = store.stats()
        print("gen %d: %d games (%.0f plies avg, %.2f M sims) in %.2f s | buffer %d games / %d examples -> %d unique | "
              "%d batches in %.2f s, loss_p %.4f loss_v %.4f" %
              (gen, a.games, prog["moves"] / a.games, prog["sims"] / 1e6, t_play, st["n_games"], st["n_examples"], n_unique,
               a.batches, t_train, lp, lv), flush=True)
        if a.eval_every and gen % a.eval_every == 0:                        # Trainer.test_agent (train.py:238-270)
            net.eval()
            t2 = time.perf_counter()
            out = []
            for agent, opp, sims in (("net", "random", 0), ("net", "uct", 100), ("zero", "uct", 200), ("net", "uct", 200)):
                s1, s2, _ = arena.play_tests(net, a.game, a.tests, agent, opp, opponent_sims=sims, device=dev, seed=gen,
                                             eval_precision=a.precision, c_puct=2.5)
                out.append("%s vs %s%s: %+.3f" % (agent, opp, sims or "", float((s1.sum() + s2.sum()) / (2 * a.tests))))
            print("        test_agent (%d tests each, %.2f s): %s" % (a.tests, time.perf_counter() - t2, " | ".join(out)), flush=True)
        if a.save:
            os.makedirs(a.save, exist_ok=True)
            torch.save(net.state_dict(), os.path.join(a.save, "%d.pth" % gen))


if __name__ == "__main__":
    main()
