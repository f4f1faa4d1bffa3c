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
  cpu_baseline   the C oracle (same algorithm, sequential playouts, batch-1 torch CPU net — what a reference
                 worker process does) playing FULL games on the host cores: all cores, one thread, and the C1 point
  opt_in_f16     (default run: --precision f32x = the product default = the reference's fp32 Net.forward grade) the SAME
                 engine, slots and weights with the fused tower in its opt-in fp16-operand mode, timed over whole steps
                 through the same 16-ticks-per-graph path; and torch_fp32: the torch module in fp32 (MIOpen), wall-bounded
  reference_precision  (only with --precision f16) the fp32-grade mode as the step-timed companion instead

`python bench.py --gpus N` without a torchrun environment starts the N ranks itself (a child `python -m
torch.distributed.run`, before this process touches the GPU) and forwards rank 0's JSON line.
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
MFMA_PEAK_TFLOPS = {"f32": 157.3, "f16": 2500.0, "bf16": 2500.0,
                    "f16x3": 2500.0 / 3}  # fp32-grade mode: three fp16 MFMAs per product
DTYPE_LABEL = {"f16x3": "f32-grade (f16x3: split-fp16 operands, 3 MFMAs per product, fp32 accumulate)"}
PROFILE_SUMMARY = os.path.join("profiles", "r4_bench_default_pmc_summary.txt")


def build_id():
    """Identity of the kernels this run times: a hash over the engine library's SOURCES (csrc/*.hip, csrc/*.h, include/*.h).  It is
    part of the workload key, so a committed PMC summary is quoted only when it was taken on the same kernels - a kernel change
    without a profile refresh yields `traffic: null`, not another build's bytes."""
    import glob
    import hashlib
    h = hashlib.sha1()
    src = os.path.join(ROOT, "alphazero-openspiel_amd", "csrc")
    for path in sorted(glob.glob(os.path.join(src, "*.hip")) + glob.glob(os.path.join(src, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))):
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()[:12]


def workload_key(game, G, S, blocks, filters, net, weights, precision, overlap, tpg):
    return "|".join(str(x) for x in (game, G, S, blocks, filters, net, weights, precision, overlap, tpg, "build:" + build_id()))


def pmc_traffic(key):
    """HBM bytes per launch from the committed rocprofv3 PMC summary (tools/profile_round.sh: separate FETCH_SIZE and
    WRITE_SIZE passes, KB per launch).  Quoted only when the summary was collected on THIS workload (its `# workload_key:`
    line); otherwise None.  Net kernels stream 16 B per lane: FETCH_SIZE doubled (gfx950 reports wide coalesced reads at
    half size, MI355X_MICROARCH.md); tick kernel: narrow accesses, uncalibrated, left as counted."""
    path = os.path.join(ROOT, PROFILE_SUMMARY)
    try:
        lines = open(path).read().splitlines()
    except OSError:
        return None, None
    if not any(ln.strip() == "# workload_key: " + key for ln in lines):
        return None, None
    kern, fetch, write = None, {}, {}
    for ln in lines:
        if ln.startswith("kernel "):
            kern = ln[7:67].strip()
        elif kern and ln.strip().startswith("FETCH_SIZE"):
            fetch[kern] = 1e3 * float(ln.rsplit("=", 1)[1])
        elif kern and ln.strip().startswith("WRITE_SIZE"):
            write[kern] = 1e3 * float(ln.rsplit("=", 1)[1])
    net = [k for k in fetch if "az_tower" in k or "az_head" in k]
    tree = [k for k in fetch if "az_advance_kernel" in k]
    if not net or not tree:
        return None, None
    return (sum(2 * fetch[k] + write.get(k, 0.0) for k in net), sum(fetch[k] + write.get(k, 0.0) for k in tree))


def net_flops_per_eval(H, W, A, n_blocks, F0=50, c_in=4):
    """F = 2HW[9*Cin*F0 + Cin*F0 + (2*n_blocks-1)*9*F0^2] + 2*F0*HW*(A+1)"""
    return 2 * H * W * (9 * c_in * F0 + c_in * F0 + (2 * n_blocks - 1) * 9 * F0 * F0) + 2 * F0 * H * W * (A + 1)


def tree_bytes_per_sim(d, a_sel, a_leaf, A, H, W, state_bytes=16, C=3):
    """B_sim = d(16*Abar+4) + 16(d+1) + 16*A_leaf + 4(A+1) + 2*S_state + 4(C+1)HW"""
    return d * (16 * a_sel + 4) + 16 * (d + 1) + 16 * a_leaf + 4 * (A + 1) + 2 * state_bytes + 4 * (C + 1) * H * W


def host_cores():
    """Cores this process may really use: the affinity mask, cut to the cgroup's CPU quota when one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_leg(n_workers, games_each, game_name, S, n_blocks, n_filters, weight_seed, budget_s, checkpoint=None):
    """n_workers fresh child processes (python -m oracle.cpu_selfplay), each playing FULL self-play games with the C oracle
    + a batch-1 torch-CPU net on one thread; they start together after a READY/GO handshake.  -> aggregate dict."""
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1", PYTHONDONTWRITEBYTECODE="1")
    cmd = [sys.executable, "-m", "oracle.cpu_selfplay", "--game", game_name, "--games", str(games_each), "--playouts", str(S),
           "--blocks", str(n_blocks), "--filters", str(n_filters), "--weight-seed", str(weight_seed), "--handshake",
           "--budget-s", str(budget_s)] + (["--checkpoint", checkpoint] if checkpoint else [])
    procs = [subprocess.Popen(cmd + ["--seed", str(100 + i)], cwd=ROOT, env=env, stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                              text=True) for i in range(n_workers)]
    try:
        for pr in procs:
            line = pr.stdout.readline()
            if line.strip() != "READY":
                raise RuntimeError("cpu_selfplay worker failed to start: %r" % line)
        t_go = time.time()
        for pr in procs:
            pr.stdin.write("go\n")
            pr.stdin.flush()
        res = [json.loads(pr.stdout.readline()) for pr in procs]
        for pr in procs:
            pr.wait()
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    wall = max(r["t_end"] for r in res) - t_go
    games = sum(r["games"] for r in res)
    sims = sum(r["sims"] for r in res)
    evals = sum(r["evals"] for r in res)
    plies = sum(r["plies"] for r in res)
    return {"games_per_s": games / wall, "sims_per_s": sims / wall, "evals_per_s": evals / wall, "games": games, "plies": plies,
            "seconds": wall, "workers": n_workers, "ms_per_eval_per_core": 1e3 * sum(r["seconds"] for r in res) / max(1, evals)}


def cpu_baseline(game_name, S, n_blocks, n_filters, weight_seed, quick=False, checkpoint=None):
    """the CPU restatement (oracle/az_oracle.c: same algorithm, strictly sequential playouts per tree, tree
    reuse, root Dirichlet, one batch-1 Net.forward per leaf through torch on the CPU - what a reference worker process does)
    on this box's host cores, FULL games: all cores (>= 64 games), one thread (8 games), and the C1 plumbing point
    (connect_four, 25 sims/move, 2-block, one thread).  Runs BEFORE this process touches the GPU; children are separate
    processes.  Every leg is bounded (a worker stops starting games after its budget; only whole games count)."""
    cores = host_cores()
    workers = max(1, min(cores, 64))
    games_each = 1 if quick else max(1, -(-64 // workers))
    many = _cpu_leg(workers, games_each, game_name, S, n_blocks, n_filters, weight_seed, budget_s=(10 if quick else 45),
                    checkpoint=checkpoint)
    one = _cpu_leg(1, 1 if quick else 8, game_name, S, n_blocks, n_filters, weight_seed, budget_s=(5 if quick else 50),
                   checkpoint=checkpoint)
    c1 = _cpu_leg(1, 2 if quick else 8, "connect_four", 25, 2, 50, weight_seed, budget_s=10)
    return {
        "value": many["games_per_s"], "unit": "games/s", "cores": workers, "kind": "port",
        "sample": "oracle/az_oracle.c play_game_self + batch-1 torch-CPU Net (1 thread per process): %d processes x FULL games, "
                  "%d games / %d plies at %d sims/move in %.1f s" % (workers, many["games"], many["plies"], S, many["seconds"]),
        "sims_per_s": many["sims_per_s"], "evals_per_s": many["evals_per_s"], "ms_per_eval_per_core": many["ms_per_eval_per_core"],
        "cpu_model": cpu_model(), "host_cpus": os.cpu_count(), "usable_cores": cores,
        "one_thread": {"value": one["games_per_s"], "sims_per_s": one["sims_per_s"], "games": one["games"],
                       "plies": one["plies"], "seconds": one["seconds"], "ms_per_eval": one["ms_per_eval_per_core"]},
        "c1_one_thread": {"workload": "connect_four, 25 sims/move, 2-block x 50, 1 CPU thread (BASELINE configs[0])",
                          "value": c1["games_per_s"], "sims_per_s": c1["sims_per_s"], "games": c1["games"],
                          "seconds": c1["seconds"]},
        # BASELINE.md section 2 (build container, 8 cores, pygames stub): the real reference's multi-process path reached
        # 0.106 games/s / 1.34 k sims/s at 400 sims (8 processes), 0.91 games/s at 25 sims (1 process, 5-block checkpoint)
        "reference_calibration": {"quoted": "constants from BASELINE.md section 2, NOT measured in this run",
                                  "where": "build container, 8 cores (BASELINE.md section 2; oracle/calibrate_reference.py)",
                                  "reference_400sims_8proc_games_per_s": 0.106, "reference_25sims_1proc_games_per_s": 0.91,
                                  # the REAL reference's in-process play_game_self (policy_fn = Net.predict, 1 thread, shipped
                                  # checkpoint) against this restatement with the same network, plies per second:
                                  "restatement_over_reference_per_ply": {"25_sims": 1.23, "400_sims": 1.00}},
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
    ap.add_argument("--dtype", default=None, choices=["f32", "f16", "bf16"], help="torch backend only (fused: see --precision)")
    ap.add_argument("--precision", default="f32x", choices=["f16", "f32x"],
                    help="fused backend: f32x = fp32-grade split-fp16 mode, 3 MFMAs per product (the product default: the reference's "
                         "Net.forward is fp32); f16 = opt-in fp16 MFMA operands")
    ap.add_argument("--companion-steps", type=int, default=2,
                    help="whole steps timed for the other-precision companion leg after the main run (0 = skip)")
    ap.add_argument("--check-every", type=int, default=128)
    ap.add_argument("--nodes-per-slot", type=int, default=0, help="node-pool capacity per slot (0 = engine default)")
    ap.add_argument("--max-sims-per-tick", type=int, default=0, help="NN-free playouts a slot may chain per tick (0 = default)")
    ap.add_argument("--chain-window-us", type=int, default=0, help="chained playouts only start this early in a launch (0 = default, <0 = off)")
    ap.add_argument("--overlap", type=int, default=1,
                    help="slot groups ticking on their own HIP streams (BASELINE configs[4]: overlapped PV-eval / tree-search "
                         "streams): a group's tree search and launch gaps run beside another group's forward")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--ticks-per-graph", type=int, default=16,
                    help="ticks captured per HIP graph (one replay = that many ticks: fewer graph-boundary bubbles; measured 1 -> 4 -> 16: "
                         "2814 -> 2873 -> 2889 games/s on one box)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", default="full", choices=["full", "quick", "off"],
                    help="full = SURVEY 8(d): all cores >= 64 full games + 1 thread 8 games + C1 point (~1.5 min); quick = 1 game per leg")
    ap.add_argument("--ref-seconds", type=float, default=10.0,
                    help="wall-time cap of the torch-fp32 (Net.forward under PyTorch-ROCm) leg that follows the timed run; 0 = skip "
                         "both companion legs")
    ap.add_argument("--tick-limit", type=int, default=0,
                    help="profiling aid (rocprofv3 --pmc passes serialise every dispatch): end the timed region after this many ticks "
                         "even if the K steps are not complete; the line then carries tick_limited = true and is NOT a bench result")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N>1 (nccl = RCCL over xGMI; gloo only to rehearse several ranks on one GPU)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N`: this process has not touched the GPU; start the N ranks as a CHILD process
        # (one rank per GPU over RCCL, rendezvous on 127.0.0.1), forward rank 0's JSON line, exit with the child's code.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    cpu = None
    if world == 1 and not args.no_cpu_baseline and args.cpu_baseline != "off":
        # host-core baseline first: child processes, before this process initialises the GPU
        ckpt = None
        if args.weights == "checkpoint":
            tag = {"connect_four": "connect_four", "breakthrough(rows=6,columns=6)": "breakthrough6"}.get(args.game)
            ckpt = os.path.join(ROOT, "tests", "golden", "checkpoint_%s.npz" % tag) if tag else None
        cpu = cpu_baseline(args.game, args.playouts, args.blocks, args.filters, args.seed, quick=args.cpu_baseline == "quick",
                           checkpoint=ckpt)
    local_rank = local_rank % max(1, torch.cuda.device_count())  # rehearsal: several ranks may share one GPU (gloo)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # AZ_DIST_FORCE=1: run the collectives of the N > 1 path in a one-rank process group (RCCL on a single-GPU box)
    distributed = world > 1 or os.environ.get("AZ_DIST_FORCE") == "1"
    if distributed:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)  # RCCL communicator bound to this rank's GPU
        else:
            dist.init_process_group(args.backend)

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
    if distributed:  # the path's first collective: every rank takes the training rank's weights (examplegenerator.py:121)
        net = net.to(device)
        azdist.broadcast_net(net, src=0)
    if args.net == "fused":
        from alphazero_openspiel_amd.fusednet import FusedNet
        args.dtype = "f16" if args.precision == "f16" else "f16x3"
        evaluator = FusedNet(net.eval(), device, max_boards=G, precision=args.precision)
    else:
        args.dtype = args.dtype or "f32"
        tdtype = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[args.dtype]
        evaluator = E.DeviceEvaluator(net, device, dtype=tdtype)

    # game capacity: warm-up + timed steps + the instrumented pass + the reference-precision legs (bounded by wall time;
    # at most ~1500 games/s) - a slot that finds no game left goes idle, which would understate those legs
    n_total = (Wm + K + 2) * G
    if world == 1:
        n_total += (max(0, args.companion_steps) + 1) * G + int(60 * max(0.0, args.ref_seconds)) + G
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
            for _ in range(max(1, args.ticks_per_graph)):
                tick_eager()
    tick = graph.replay if graph is not None else tick_eager
    tpg = max(1, args.ticks_per_graph) if graph is not None else 1
    tpg_main = tpg
    if args.check_every % tpg:
        raise SystemExit("--check-every must be a multiple of --ticks-per-graph")
    sync_groups = lambda: None
    if args.overlap > 1:  # k slot groups, each [advance_slots, forward] on its own stream / graph
        if args.net != "fused":
            raise SystemExit("--overlap needs --net fused")
        group_nets = [FusedNet(net.eval(), device, max_boards=n, precision=args.precision) for _, n in E.slot_groups(G, args.overlap)]
        tk = E.OverlappedTicker(eng, group_nets, args.overlap, use_graph=not args.no_graph, io=(obs, pri, val))
        tick, sync_groups, graph, tpg = tk.tick, tk.synchronize, (tk.graphs or None), 1  # (one tick per group graph)

    def barrier():
        if distributed:
            if args.backend == "nccl":
                dist.barrier(device_ids=[local_rank])
            else:
                dist.barrier()

    def run_until(n_done):
        ticks = 0
        while True:
            for _ in range(args.check_every // tpg):
                tick()
            ticks += args.check_every
            sync_groups()
            if eng.games_done() >= n_done or (args.tick_limit and ticks >= args.tick_limit):
                return eng.progress(), ticks

    # ---- warm-up: W steps (W*G games completed) -------------------------------------------------
    if Wm > 0:
        p0, _ = run_until(Wm * G)
    else:
        for _ in range(args.check_every):
            tick()
        sync_groups()
        p0 = eng.progress()
    barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    # ---- timed: exactly K steps --------------------------------------------------------------------
    p1, ticks = run_until((Wm + K) * G)
    t_ag0 = time.perf_counter()
    allgather_ms = None
    if distributed:  # generation-end exchange of the finished-game records (the path's only collective)
        buf = eng.export_device()  # packed records, device resident
        gathered = azdist.all_gather_device_exports(buf)  # RCCL all-gather over xGMI (nccl backend)
        torch.cuda.synchronize(device)
        assert gathered.numel() == world * buf.numel()
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
    if distributed:
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
    net_tflops = evals_tick * f_eval / t_net / 1e12  # evaluated leaves only: slots without a request hold stale boards
    e2e_tflops = evals_all / world / dt_all * f_eval / 1e12  # per GPU, over the whole timed region (gaps + tick kernel included)
    tree_gbs = sims_tick * b_sim / t_tree / 1e9
    peak = MFMA_PEAK_TFLOPS[args.dtype]
    # MFMA work actually ISSUED (padding included) against the 2.5 PFLOP/s fp16 pipe: v_mfma_f32_16x16x32_f16 = 16384 FLOP
    issued_mfma = evaluator.issued_mfma_per_board() if args.net == "fused" else None
    issued_frac = (issued_mfma * 16384.0 * evals_tick / t_net / 1e12 / MFMA_PEAK_TFLOPS["f16"]) if issued_mfma else None

    # HBM bytes per launch: rocprofv3 PMC counters cannot be collected from inside this process, so they are read from the
    # committed summary of the profiled command and quoted only when this run IS that workload (pmc_traffic)
    traffic_net = traffic_tree = None
    if args.net == "fused":
        traffic_net, traffic_tree = pmc_traffic(workload_key(game.name, G, S, args.blocks, args.filters, args.net, args.weights,
                                                             args.precision, args.overlap, tpg))

    # ---- companion legs: the SAME engine, slots, games in flight and net weights under another evaluator -----------------
    #   steps_leg  whole steps (n_steps x G games completed), the same tpg-ticks-per-graph replay path as the main run:
    #              the fused tower at the OTHER precision (main f32x -> opt_in_f16; main f16 -> reference_precision)
    #   wall_leg   bounded wall time, eager: the torch module in fp32 under PyTorch-ROCm (MIOpen) = what the reference's
    #              handle_gpu would run on this device (examplegenerator.py:72-77)
    def forward_ms(ev, n=9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ms = []
        for _ in range(n):
            e0.record()
            ev(obs, pri, val)
            e1.record()
            e1.synchronize()
            ms.append(e0.elapsed_time(e1))
        return float(np.median(ms))

    def leg_result(q0, q1, dtr, n_ticks, ms):
        return {"value": (q1["games_done"] - q0["games_done"]) / dtr, "unit": "games/s",
                "sims_per_s": (q1["sims"] - q0["sims"]) / dtr, "evals_per_s": (q1["evals"] - q0["evals"]) / dtr,
                "games_counted": q1["games_done"] - q0["games_done"], "ticks": n_ticks, "seconds": dtr, "ms_per_launch": ms,
                "tflops": (q1["evals"] - q0["evals"]) / max(1, n_ticks) * f_eval / (ms * 1e-3) / 1e12}

    def steps_leg(ev, n_steps):
        for _ in range(2):  # workspace allocation outside capture
            eng.advance(pri, val, obs)
            ev(obs, pri, val)
        torch.cuda.synchronize(device)
        g = None
        if not args.no_graph:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(tpg_main):
                    eng.advance(pri, val, obs)
                    ev(obs, pri, val)
        ms = forward_ms(ev)
        q0 = eng.progress()
        target = q0["games_done"] + n_steps * G
        torch.cuda.synchronize(device)
        tr0 = time.perf_counter()
        n_ticks = 0
        while True:
            for _ in range(args.check_every // tpg_main):
                if g is not None:
                    g.replay()
                else:
                    eng.advance(pri, val, obs)
                    ev(obs, pri, val)
            n_ticks += args.check_every
            if eng.games_done() >= target:
                break
        torch.cuda.synchronize(device)
        dtr = time.perf_counter() - tr0
        r = leg_result(q0, eng.progress(), dtr, n_ticks, ms)
        r.update({"steps": n_steps, "ms_per_step": 1e3 * dtr / n_steps, "ticks_per_graph": tpg_main if g is not None else 1})
        return r

    def wall_leg(ev, seconds):
        for _ in range(2):  # kernel selection / workspace allocation outside the window
            eng.advance(pri, val, obs)
            ev(obs, pri, val)
        torch.cuda.synchronize(device)
        ms = forward_ms(ev, 5)
        q0 = eng.progress()
        tr0 = time.perf_counter()
        n_ticks = 0
        while time.perf_counter() - tr0 < seconds:
            for _ in range(8):
                eng.advance(pri, val, obs)
                ev(obs, pri, val)
                n_ticks += 1
            torch.cuda.synchronize(device)
        dtr = time.perf_counter() - tr0
        return leg_result(q0, eng.progress(), dtr, n_ticks, ms)

    companion = None
    if world == 1 and args.net == "fused" and args.overlap == 1 and args.ref_seconds > 0:
        sync_groups()
        other = "f16" if args.precision == "f32x" else "f32x"
        if args.companion_steps > 0:
            ev_o = FusedNet(net.eval(), device, max_boards=G, precision=other)
            companion = steps_leg(ev_o, args.companion_steps)
            odt = "f16" if other == "f16" else "f16x3"
            companion.update({"dtype": DTYPE_LABEL.get(odt, odt), "evaluator": ev_o.kernel_label() + ", same engine / slots / weights",
                              "peak_tflops": MFMA_PEAK_TFLOPS[odt], "frac": companion["tflops"] / MFMA_PEAK_TFLOPS[odt],
                              "issued_mfma_per_board": ev_o.issued_mfma_per_board(),
                              "frac_issued": ev_o.issued_mfma_per_board() * 16384.0 * (companion["tflops"] / f_eval) / MFMA_PEAK_TFLOPS["f16"]})
            del ev_o
        ev32 = E.DeviceEvaluator(net, device, dtype=torch.float32)
        t32 = wall_leg(ev32, args.ref_seconds)
        t32.update({"dtype": "f32", "evaluator": "Net.forward fp32 under PyTorch-ROCm (MIOpen), eager",
                    "peak_tflops": MFMA_PEAK_TFLOPS["f32"]})
        if companion is None:
            companion = {}
        companion["torch_fp32"] = t32

    if rank == 0:
        plies_per_game = moves_all / max(1.0, games_all)
        out = {
            "metric": "self-play games/sec", "value": games_all / dt_all, "unit": "games/s",
            "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": 1e3 * dt_all / K,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE_LABEL.get(args.dtype, args.dtype), "data": "synthetic",
            "config": {"workload": "%s, %d sims/move, %d-block x %d-filter ResNet, %d concurrent games per GPU%s"
                                   % (game.name, S, args.blocks, args.filters, G,
                                      "" if args.overlap < 2 else " in %d slot groups on %d HIP streams" % (args.overlap, args.overlap)),
                       "overlap": args.overlap,
                       "weights": ("random-init (torch.manual_seed), eval-mode BN" if args.weights == "random" else
                                   "the reference's shipped checkpoint (5-block x 50)"), "net_backend": args.net,
                       "net_precision": (("fp32-grade: every operand a pair of fp16 numbers (hi + lo/2048), three MFMAs per product, fp32 "
                                          "accumulate / residual stream / epilogues (ExampleGenerator default eval_precision='f32x'; the "
                                          "reference's Net.forward is fp32, network.py:48-64)") if args.precision == "f32x" else
                                         "fp16 MFMA operands, fp32 accumulate / residual stream (opt-in eval_precision='f16')")
                                        if args.net == "fused" else args.dtype,
                       "tree_dtype": "f64", "c_puct": 2.5, "temperature": 1.0, "dirichlet_alpha": 0.3,
                       "parallelism": "games sharded over %d GPU(s), no collective inside the search" % world,
                       "hip_graph": graph is not None, "ticks_per_graph": tpg,
                       "workload_key": workload_key(game.name, G, S, args.blocks, args.filters, args.net, args.weights, args.precision,
                                                    args.overlap, tpg)},
            "sims_per_s": sims_all / dt_all, "evals_per_s": evals_all / dt_all,
            "games_timed": games_all, "ticks_timed_rank0": ticks, "tick_limited": bool(args.tick_limit),
            "mean_plies_per_game": plies_per_game, "mean_select_depth": d_mean, "mean_children_scanned": a_sel,
            "terminal_hit_fraction": (p1["terminal_hits"] - p0["terminal_hits"]) / max(1, sims),
            "allgather_ms": allgather_ms, "compactions": p1["compactions"] - p0["compactions"],
            "engine_hbm_gb": eng.sizes.device_bytes / 1e9,
            "roofline": {"bound": "mfma", "kernel": (evaluator.kernel_label() if args.net == "fused" else "torch Net.forward (MIOpen)")
                                   + ", %d boards/launch" % G,
                         "achieved": net_tflops, "peak": peak, "unit": "TFLOP/s", "frac": net_tflops / peak,
                         "peak_note": ("dense fp16 MFMA peak 2500 TFLOP/s / 3 MFMAs per product" if args.dtype == "f16x3" else
                                       "dense MFMA peak for the dtype (MI355X_MICROARCH.md)"),
                         # the same achieved rate against the guide's RAW peaks, whatever `peak` was derived from
                         "frac_of_fp16_peak": net_tflops / MFMA_PEAK_TFLOPS["f16"], "x_fp32_mfma_peak": net_tflops / MFMA_PEAK_TFLOPS["f32"],
                         "frac_issued": issued_frac, "issued_mfma_per_board": issued_mfma,
                         "frac_end_to_end": e2e_tflops / peak, "end_to_end_tflops": e2e_tflops,
                         "traffic": traffic_net, "traffic_unit": "bytes/launch (PMC, %s)" % PROFILE_SUMMARY,
                         "flops_per_eval": f_eval, "ms_per_launch": 1e3 * t_net,
                         "batch_fill": evals_tick / G},
            "roofline_tree": {"bound": "hbm", "kernel": "az_advance_kernel (playouts + move step)",
                              "achieved": tree_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": tree_gbs / HBM_PEAK_GBS, "traffic": traffic_tree, "bytes_per_sim": b_sim,
                              "sims_per_launch": sims_tick, "ms_per_launch": 1e3 * t_tree},
        }
        if companion is not None:
            out["opt_in_f16" if args.precision == "f32x" else "reference_precision"] = companion
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out))
    eng.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
