#!/usr/bin/env python3
"""Wall clock of ONE reference-facing call, host format included: ExampleGenerator.generate_examples(n) (self-play on the
device + one device-to-host copy + building the reference's Python lists) next to generate_into(replay, n) (records stay in
HBM).  bench.py times the device path alone; this is the number a caller of the drop-in sees.

    python tools/generation_wallclock.py [--games 4096] [--playouts 400] [--blocks 10] [--precision f16|f32x]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphazero_openspiel_amd import engine as E, replay  # noqa: E402
from alphazero_openspiel_amd.examplegenerator import ExampleGenerator  # noqa: E402
from alphazero_openspiel_amd.network import Net  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--game", default="connect_four", choices=["connect_four"])
    ap.add_argument("--games", type=int, default=4096)
    ap.add_argument("--playouts", type=int, default=400)
    ap.add_argument("--blocks", type=int, default=10)
    ap.add_argument("--precision", default="f16")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    gen0 = ExampleGenerator(Net([3, 6, 7], 7, n_blocks=a.blocks), a.game, dev,
                            n_playouts=a.playouts, eval_precision=a.precision, seed=1)
    out = {"game": a.game, "games": a.games, "n_playouts": a.playouts, "blocks": a.blocks, "precision": a.precision}
    gen0.generate_examples(min(a.games, 256))  # warm-up: library load, graph capture paths
    torch.cuda.synchronize()
    t = time.perf_counter()
    games = gen0.generate_examples(a.games)
    t_total = time.perf_counter() - t
    n_ex = sum(len(g) for g in games)
    # the same call split: device part (self-play + packed export), then the host format
    t = time.perf_counter()
    gathered, nbytes, n_local, world, (mp, mc) = gen0._play_and_gather(a.games)
    torch.cuda.synchronize()
    t_dev = time.perf_counter() - t
    t = time.perf_counter()
    host = gathered.cpu().numpy()
    t_copy = time.perf_counter() - t
    t = time.perf_counter()
    ex = E.unpack_device_export(host[:nbytes], n_local, mp, mc)
    lists = E.examples_from_export(gen0.game, ex)
    t_lists = time.perf_counter() - t
    store = replay.DeviceReplay(gen0.game, max_games=a.games, device=dev)
    t = time.perf_counter()
    gen0.generate_into(store, a.games)
    torch.cuda.synchronize()
    t_into = time.perf_counter() - t
    out.update(examples=n_ex, generate_examples_s=round(t_total, 3), generate_examples_games_per_s=round(a.games / t_total, 1),
               device_part_s=round(t_dev, 3), d2h_copy_s=round(t_copy, 4), d2h_bytes=int(gathered.numel()),
               python_lists_s=round(t_lists, 3), examples_per_s_host_format=round(sum(len(g) for g in lists) / t_lists),
               generate_into_s=round(t_into, 3), generate_into_games_per_s=round(a.games / t_into, 1))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
