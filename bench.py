#!/usr/bin/env python3
"""bench.py — self-play games/sec of the HIP engine (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload at N=1 = BASELINE.json configs[1]: connect_four, 400 sims/move, 10-block x 50-filter ResNet,
4096 concurrent games on one MI355X, random-initialised weights (no 10-block checkpoint exists),
c_puct 2.5, temperature 1, Dirichlet(0.3) root noise, tree reuse, backup "on-policy".

A STEP = one generation-equivalent: G (= --slots) self-play games completed, with device-side slot
refill so the request batch stays full.  Warm-up = W steps (brings the slots out of lock-step), then
exactly K steps are timed between barrier + synchronize pairs; value = games completed in the timed
region / max-over-ranks time, summed over ranks ("weak" scaling: G games resident per GPU).
With N > 1 each rank plays its own shard; the one collective of the path — the all-gather of the
finished-game records at generation end — is inside the timed region and also reported separately.

Extra objects on the JSON line:
  roofline       dominant kernel(s) of a tick = the PV-net forward (MFMA bound): algorithmic FLOPs per
                 evaluated leaf x leaves per tick / HIP-event time of the forward
  roofline_tree  the search kernels (HBM bound, latency-limited): SURVEY.md §8(d) bytes per sim with the
                 measured mean depth / children x sims per tick / HIP-event time of az_engine_advance
  cpu_baseline   the C oracle (same algorithm, sequential playouts, batch-1 torch CPU net — what the
                 reference does in-process) on 1 host thread over a bounded number of moves
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_PEAK_TFLOPS = {"f32": 157.3, "f16": 2500.0, "bf16": 2500.0}


def net_flops_per_eval(H, W, A, n_blocks, F0=50, c_in=4):
    """F = 2HW[9*Cin*F0 + Cin*F0 + (2*n_blocks-1)*9*F0^2] + 2*F0*HW*(A+1)"""
    return 2 * H * W * (9 * c_in * F0 + c_in * F0 + (2 * n_blocks - 1) * 9 * F0 * F0) + 2 * F0 * H * W * (A + 1)


def tree_bytes_per_sim(d, a_sel, a_leaf, A, H, W, state_bytes=16, C=3):
    """B_sim = d(16*Abar+4) + 16(d+1) + 16*A_leaf + 4(A+1) + 2*S_state + 4(C+1)HW"""
    return d * (16 * a_sel + 4) + 16 * (d + 1) + 16 * a_leaf + 4 * (A + 1) + 2 * state_bytes + 4 * (C + 1) * H * W


def cpu_baseline(game_name, S, n_blocks, n_filters, mean_evals_per_game, budget_s=15.0):
    """The oracle (C restatement) with the same Net on the host, 1 thread, batch-1 per leaf."""
    from oracle import binding as orc
    from alphazero_openspiel_amd.games import Game
    from alphazero_openspiel_amd.network import Net

    torch.set_num_threads(1)
    g = Game(game_name)
    A = g.num_distinct_actions()
    shape = g.information_state_normalized_vector_shape()
    torch.manual_seed(0)
    net = Net(shape, A, n_blocks=n_blocks, n_filters=n_filters).eval()

    def policy(board):
        with torch.no_grad():
            p, v = net(torch.from_numpy(board.reshape(1, 4, g.rows, g.cols)).float())
        return p[0].double().numpy(), float(v)

    b0 = np.zeros(4 * g.rows * g.cols)
    policy(b0)
    t = time.perf_counter()
    for _ in range(20):
        policy(b0)
    t_eval = (time.perf_counter() - t) / 20
    moves = int(max(1, min(12, budget_s / ((S + 1) * t_eval))))
    t = time.perf_counter()
    out = orc.play_game_self(policy, game_name, n_playouts=S, seed=1, max_moves=moves)
    dt = time.perf_counter() - t
    c = out["counters"]
    evals_per_s = c["evals"] / dt
    return {
        "value": evals_per_s / mean_evals_per_game if mean_evals_per_game else None,
        "unit": "games/s", "cores": 1, "kind": "port",
        "sample": "oracle/az_oracle.c play_game_self, first %d moves of one game at %d sims/move (%d playouts, %d "
                  "batch-1 torch-CPU net evals in %.1f s); games/s = CPU evals/s / the GPU run's mean net evals per game"
                  % (len(out["actions"]), S, c["sims"], c["evals"], dt),
        "sims_per_s": c["sims"] / dt, "evals_per_s": evals_per_s, "ms_per_eval": 1e3 * t_eval,
        "host_cpus": os.cpu_count(),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--game", default="connect_four")
    ap.add_argument("--slots", type=int, default=4096)
    ap.add_argument("--playouts", type=int, default=400)
    ap.add_argument("--blocks", type=int, default=10)
    ap.add_argument("--filters", type=int, default=50)
    ap.add_argument("--weights", default="random", choices=["random", "checkpoint"],
                    help="checkpoint = the reference's shipped 5-block x 50 nets (tests/golden/checkpoint_*.npz; connect_four and "
                         "breakthrough 6x6 only): realistic priors, game lengths and tree shapes; overrides --blocks/--filters")
    ap.add_argument("--net", default="fused", choices=["fused", "torch"],
                    help="fused = csrc/az_net.hip MFMA tower (fp16 operands, fp32 accumulate); torch = nn.Module under PyTorch-ROCm")
    ap.add_argument("--dtype", default=None, choices=["f32", "f16", "bf16"], help="torch backend only (fused is f16)")
    ap.add_argument("--check-every", type=int, default=128)
    ap.add_argument("--nodes-per-slot", type=int, default=0, help="node-pool capacity per slot (0 = engine default)")
    ap.add_argument("--max-sims-per-tick", type=int, default=0, help="NN-free playouts a slot may chain per tick (0 = default)")
    ap.add_argument("--chain-window-us", type=int, default=0, help="chained playouts only start this early in a launch (0 = default, <0 = off)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N>1 (nccl = RCCL over xGMI; gloo only to rehearse several ranks on one GPU)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(args.backend)
    local_rank = local_rank % max(1, torch.cuda.device_count())  # rehearsal: several ranks may share one GPU (gloo)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    from alphazero_openspiel_amd import distributed as azdist
    from alphazero_openspiel_amd import engine as E
    from alphazero_openspiel_amd.games import Game
    from alphazero_openspiel_amd.network import Net

    game = Game(args.game)
    A = game.num_distinct_actions()
    H, Wd = game.rows, game.cols
    G, S, K, Wm = args.slots, args.playouts, args.steps, args.warmup
    torch.manual_seed(args.seed)
    if args.weights == "checkpoint":
        from alphazero_openspiel_amd.network import load_npz_checkpoint
        tag = {"connect_four": "connect_four", "breakthrough(rows=6,columns=6)": "breakthrough6"}[game.name]
        net = load_npz_checkpoint(os.path.join(ROOT, "tests", "golden", "checkpoint_%s.npz" % tag),
                                  game.information_state_normalized_vector_shape(), A)
        args.blocks, args.filters = net.n_blocks, net.n_filts
    else:
        net = Net(game.information_state_normalized_vector_shape(), A, n_blocks=args.blocks, n_filters=args.filters)
    if args.net == "fused":
        from alphazero_openspiel_amd.fusednet import FusedNet
        args.dtype = "f16"
        evaluator = FusedNet(net.eval(), device, max_boards=G)
    else:
        args.dtype = args.dtype or "f32"
        tdtype = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[args.dtype]
        evaluator = E.DeviceEvaluator(net, device, dtype=tdtype)

    n_total = (Wm + K + 2) * G
    eng = E.SelfPlayEngine(game, G, n_playouts=S, max_games=n_total, device=device, seed=args.seed + 7919 * rank,
                           nodes_per_slot=args.nodes_per_slot, max_sims_per_tick=args.max_sims_per_tick,
                           chain_window_us=args.chain_window_us)
    eng.reset(n_total)
    obs, pri, val = eng.alloc_io()

    def tick_eager():
        eng.advance(pri, val, obs)
        evaluator(obs, pri, val)

    graph = None
    if not args.no_graph:
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            for _ in range(3):
                tick_eager()
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize(device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            tick_eager()
    tick = graph.replay if graph is not None else tick_eager

    def barrier():
        if world > 1:
            dist.barrier()

    def run_until(n_done):
        ticks = 0
        while True:
            for _ in range(args.check_every):
                tick()
            ticks += args.check_every
            if eng.games_done() >= n_done:
                return eng.progress(), ticks

    # ---- warm-up: W steps (W*G games completed) -------------------------------------------------
    if Wm > 0:
        p0, _ = run_until(Wm * G)
    else:
        for _ in range(args.check_every):
            tick()
        p0 = eng.progress()
    barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    # ---- timed: exactly K steps --------------------------------------------------------------------
    p1, ticks = run_until((Wm + K) * G)
    t_ag0 = time.perf_counter()
    allgather_ms = None
    if world > 1:  # generation-end exchange of the finished-game records (the path's only collective)
        packed = azdist.pack_export(eng.export())
        exports = azdist.all_gather_exports(packed, device)
        assert len(exports) == world
        allgather_ms = 1e3 * (time.perf_counter() - t_ag0)
    torch.cuda.synchronize(device)
    barrier()
    dt = time.perf_counter() - t0

    games = p1["games_done"] - p0["games_done"]
    sims = p1["sims"] - p0["sims"]
    evals = p1["evals"] - p0["evals"]
    moves = p1["moves"] - p0["moves"]
    tot = torch.tensor([dt, float(games), float(sims), float(evals), float(moves)], dtype=torch.float64,
                       device=device if args.backend == "nccl" else "cpu")
    if world > 1:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        tot[0] = mx[0]
    dt_all, games_all, sims_all, evals_all, moves_all = [float(x) for x in tot.tolist()]

    # ---- instrumented eager pass: HIP events on the launch stream ------------------------------------
    n_probe = 24
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(n_probe)]
    pa = eng.progress()
    for i in range(n_probe):
        ev[i][0].record()
        eng.advance(pri, val, obs)
        ev[i][1].record()
        evaluator(obs, pri, val)
        ev[i][2].record()
    torch.cuda.synchronize(device)
    pb = eng.progress()
    t_tree = float(np.median([e[0].elapsed_time(e[1]) for e in ev])) * 1e-3
    t_net = float(np.median([e[1].elapsed_time(e[2]) for e in ev])) * 1e-3
    sims_tick = (pb["sims"] - pa["sims"]) / n_probe
    evals_tick = (pb["evals"] - pa["evals"]) / n_probe
    d_mean = (p1["sum_depth"] - p0["sum_depth"]) / max(1, sims)
    a_sel = (p1["sum_children"] - p0["sum_children"]) / max(1, p1["sum_depth"] - p0["sum_depth"])
    a_leaf = (p1["nodes_allocated"] - p0["nodes_allocated"]) / max(1, evals)
    b_sim = tree_bytes_per_sim(d_mean, a_sel, a_leaf, A, H, Wd)
    f_eval = net_flops_per_eval(H, Wd, A, args.blocks, args.filters)
    net_tflops = G * f_eval / t_net / 1e12  # the forward runs over all G slots every tick
    tree_gbs = sims_tick * b_sim / t_tree / 1e9
    peak = MFMA_PEAK_TFLOPS[args.dtype]

    # HBM bytes per launch measured with rocprofv3 PMC counters on this exact command (separate FETCH_SIZE and
    # WRITE_SIZE passes, profiles/r1_hbm_traffic_pmc.txt).  They cannot be collected from inside this process, so
    # they are quoted only when the workload is the profiled one.  Net: FETCH_SIZE doubled (wide 16-B/lane streams
    # are reported at half size on gfx950, MI355X_MICROARCH.md); tree: uncorrected (narrow accesses, uncalibrated).
    traffic_net = traffic_tree = None
    if (game.name, G, S, args.blocks, args.filters, args.net, args.weights) == ("connect_four", 4096, 400, 10, 50,
                                                                                "fused", "random"):
        traffic_net = 2 * (5257.5e3 + 11125.5e3) + 21547.2e3 + 128.0e3
        traffic_tree = 4877.7e3 + 5802.6e3

    if rank == 0:
        plies_per_game = moves_all / max(1.0, games_all)
        out = {
            "metric": "self-play games/sec", "value": games_all / dt_all, "unit": "games/s",
            "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": 1e3 * dt_all / K,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "%s, %d sims/move, %d-block x %d-filter ResNet, %d concurrent games per GPU"
                                   % (game.name, S, args.blocks, args.filters, G),
                       "weights": ("random-init (torch.manual_seed), eval-mode BN" if args.weights == "random" else
                                   "the reference's shipped checkpoint (5-block x 50)"), "net_backend": args.net,
                       "tree_dtype": "f64", "c_puct": 2.5, "temperature": 1.0, "dirichlet_alpha": 0.3,
                       "parallelism": "games sharded over %d GPU(s), no collective inside the search" % world,
                       "hip_graph": graph is not None},
            "sims_per_s": sims_all / dt_all, "evals_per_s": evals_all / dt_all,
            "games_timed": games_all, "ticks_timed_rank0": ticks,
            "mean_plies_per_game": plies_per_game, "mean_select_depth": d_mean, "mean_children_scanned": a_sel,
            "terminal_hit_fraction": (p1["terminal_hits"] - p0["terminal_hits"]) / max(1, sims),
            "allgather_ms": allgather_ms, "compactions": p1["compactions"] - p0["compactions"],
            "engine_hbm_gb": eng.sizes.device_bytes / 1e9,
            "roofline": {"bound": "mfma", "kernel": ("az_tower_kernel + az_head_kernel" if args.net == "fused" else "torch Net.forward (MIOpen)")
                                   + ", %d boards/launch" % G,
                         "achieved": net_tflops, "peak": peak, "unit": "TFLOP/s", "frac": net_tflops / peak,
                         "traffic": traffic_net, "traffic_unit": "bytes/launch (PMC, profiles/r1_hbm_traffic_pmc.txt)",
                         "flops_per_eval": f_eval, "ms_per_launch": 1e3 * t_net,
                         "batch_fill": evals_tick / G},
            "roofline_tree": {"bound": "hbm", "kernel": "az_advance_kernel (playouts + move step)",
                              "achieved": tree_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": tree_gbs / HBM_PEAK_GBS, "traffic": traffic_tree, "bytes_per_sim": b_sim,
                              "sims_per_launch": sims_tick, "ms_per_launch": 1e3 * t_tree},
        }
        if world == 1 and not args.no_cpu_baseline:
            evals_per_game = evals_all / max(1.0, games_all)
            out["cpu_baseline"] = cpu_baseline(game.name, S, args.blocks, args.filters, evals_per_game)
        print(json.dumps(out))
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
