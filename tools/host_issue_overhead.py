#!/usr/bin/env python3
"""What ONE host thread spends keeping a self-play engine fed, against the GPU time it enqueues - the quantity an 8-GPU run of
`n_pools` engines in one process (examplegenerator.py:140-162; engine.run_selfplay_pools) or of eight ranks rests on:

  * host time to enqueue one captured graph of 16 ticks (graph.replay(): returns when the launch is queued), and the GPU time
    that replay runs;
  * host time of one az_engine_poll (the only device-to-host traffic of the tick loop: two words, after a stream sync);
  * k = 1, 2, 4, 8 engines on THIS one device ticked by one thread (run_selfplay_pools): the aggregate rate must stay the
    one-engine rate (the engines share the GPU) - a drop would be host issue cost, which is what eight devices would then see.

    python tools/host_issue_overhead.py > profiles/r4_host_issue_overhead.json
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphazero_openspiel_amd import engine as E, fusednet  # noqa: E402
from alphazero_openspiel_amd.network import Net  # noqa: E402

GAME, S, BLOCKS, G = "connect_four", 400, 10, 4096


def one_engine_replay_cost():
    torch.manual_seed(0)
    net = Net([3, 6, 7], 7, n_blocks=BLOCKS, n_filters=50).eval()
    fn = fusednet.FusedNet(net, "cuda:0", max_boards=G, precision="f32x")
    eng = E.SelfPlayEngine(GAME, G, n_playouts=S, max_games=10 ** 6, seed=1, device=0)
    eng.reset(10 ** 6)
    obs, pri, val = eng.alloc_io()
    side = torch.cuda.Stream(0)
    with torch.cuda.stream(side):
        for _ in range(3):
            eng.advance(pri, val, obs)
            fn(obs, pri, val)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(16):
            eng.advance(pri, val, obs)
            fn(obs, pri, val)
    for _ in range(20):  # warm: games spread over their plies
        graph.replay()
    torch.cuda.synchronize()
    host, n = [], 200
    t_all = time.perf_counter()
    for _ in range(n):
        t = time.perf_counter()
        graph.replay()
        host.append(time.perf_counter() - t)
    t_enq = time.perf_counter() - t_all
    torch.cuda.synchronize()
    t_gpu = time.perf_counter() - t_all
    polls = []
    for _ in range(200):
        t = time.perf_counter()
        eng.games_done()
        polls.append(time.perf_counter() - t)
    eng.close()
    fn.close()
    host.sort()
    polls.sort()
    return {"ticks_per_graph": 16, "replays": n,
            "host_us_per_replay_median": 1e6 * host[n // 2], "host_us_per_replay_p99": 1e6 * host[int(0.99 * n)],
            "host_s_enqueue_all": t_enq, "gpu_ms_per_replay": 1e3 * t_gpu / n,
            "host_fraction_of_gpu_time": (sum(host) / n) / (t_gpu / n),
            "poll_us_median_idle_stream": 1e6 * polls[100]}


def pools_on_one_device(k, games_each):
    torch.manual_seed(0)
    net = Net([3, 6, 7], 7, n_blocks=BLOCKS, n_filters=50).eval()
    slots = G // k
    engines = [E.SelfPlayEngine(GAME, slots, n_playouts=S, max_games=games_each, seed=1 + 7919 * i, device=0) for i in range(k)]
    evs = [fusednet.FusedNet(net, "cuda:0", max_boards=slots, precision="f32x") for _ in range(k)]
    torch.cuda.synchronize()
    t = time.perf_counter()
    progs = E.run_selfplay_pools(engines, evs, games_each, use_graph=True, check_every=128, compact_tail=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    for e in engines:
        e.close()
    for ev in evs:
        ev.close()
    done = sum(p["games_done"] for p in progs)
    return {"pools": k, "slots_each": slots, "games": done, "seconds": dt, "games_per_s": done / dt,
            "ticks_each": [p["ticks"] for p in progs]}


def main():
    out = {"workload": "%s, %d sims/move, %d-block x 50, f32x, %d slots on one MI355X" % (GAME, S, BLOCKS, G)}
    out["one_engine"] = one_engine_replay_cost()
    # the same 4096 boards in flight split over k engines of one process on ONE device: the GPU work is the same (smaller towers
    # run less efficiently: 1024 boards = one round of the kernel), the host thread enqueues k graphs per pass
    out["pools_on_one_device"] = [pools_on_one_device(k, 8192 // k) for k in (1, 2, 4, 8)]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
