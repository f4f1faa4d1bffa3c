"""`ExampleGenerator` with the reference's signature (examplegenerator.py:80-175), on the HIP engine.

Reference: `n_pools` pools, each = one busy-polling `handle_gpu` process + `n_processes` worker
processes playing one game at a time, a pickled batch-1 board per leaf over a Pipe
(examplegenerator.py:39-77,106-138).  Here: one process per GPU, all games of the shard resident on
the device as engine slots, one `net.forward` per tick over every outstanding leaf; across GPUs the
games shard over `torch.distributed` ranks and the examples are all-gathered once at generation end
(RCCL over xGMI; nothing is exchanged during the search).
"""
import copy

import numpy as np
import torch

from .engine import (DeviceEvaluator, EngineError, SelfPlayEngine, examples_from_export, run_selfplay,
                     slot_groups, unpack_device_export)
from .games import Game
from . import distributed as azdist

_ENGINE_KW = ("n_playouts", "c_puct", "temperature", "dirichlet_ratio", "use_dirichlet", "keep_search_tree",
              "backup", "use_puct", "num_probabilistic_actions")


def _sum_progress(progs):
    """The pools' progress dicts as one: counters summed, fault flags OR-ed, the per-pool dicts kept under "pools"."""
    out = {}
    for k in progs[0]:
        vals = [p[k] for p in progs]
        if k == "error_flags":
            v = 0
            for x in vals:
                v |= int(x)
            out[k] = v
        elif all(isinstance(x, (int, float)) and not isinstance(x, bool) for x in vals):
            out[k] = max(vals) if k in ("ticks", "tail_compactions") else sum(vals)
        else:
            out[k] = vals[0]
    out["pools"] = list(progs)
    return out


class ExampleGenerator:
    def __init__(self, net, game_name, device, n_pools=1, n_processes=1, **kwargs):
        self.net2 = copy.deepcopy(kwargs["net2"]) if kwargs.get("net2") is not None else None  # examplegenerator.py:88-90
        self.is_test = bool(kwargs.get("is_test", False))
        self.generate_statistics = bool(kwargs.get("generate_statistics", False))
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise EngineError("ExampleGenerator needs a HIP device: self-play runs in HIP kernels, there is no "
                              "CPU fallback (got device=%s)" % (device,))
        self.net = copy.deepcopy(net)  # examplegenerator.py:86: a frozen copy of the current net
        self.game_name = game_name
        self.game = Game(game_name)
        # n_pools = "amount of GPUs to utilize" (train.py:32): with several HIP devices visible to ONE process (no torch.distributed)
        # pool i is an engine of its own on the device the reference would pick (examplegenerator.py:144-150) and plays
        # int(n_games / n_pools) games; on one device the pools collapse into one engine that plays the same COUNT.
        # pool_devices=[...] names the devices explicitly (also: several pools on one device, which is how the path is tested).
        self.n_pools, self.n_processes = int(n_pools), n_processes  # (no worker processes here: module doc)
        self.pool_devices = kwargs.get("pool_devices")
        self.kwargs = kwargs
        # engine extensions (not reference keywords)
        self.n_slots = kwargs.get("n_slots")           # concurrent games per GPU; default min(n_games, 4096)
        self.seed = int(kwargs.get("seed", np.random.randint(0, 2 ** 31 - 1)))
        self.eval_backend = kwargs.get("eval_backend", "fused")   # "fused" (csrc/az_net.hip) | "torch"
        # fused backend: "f32x" = fp32-grade (split-fp16 operands, the reference's Net.forward is fp32: network.py:48-64);
        # "f16" = fp16 operands, ~2.5x the throughput, opt-in (tolerances: tests/test_precision_search_gpu.py)
        self.eval_precision = kwargs.get("eval_precision", "f32x")
        self.eval_dtype = kwargs.get("eval_dtype", torch.float32)   # torch backend only
        self.use_graph = bool(kwargs.get("use_graph", True))
        self.overlap = int(kwargs.get("overlap", 1))  # slot groups ticking on their own HIP streams (engine.run_selfplay)
        self.last_progress = None
        self._generation = 0

    def _engine_kwargs(self):
        return {k: self.kwargs[k] for k in _ENGINE_KW if k in self.kwargs}

    def _pool_device_list(self):
        """Devices of the reference's pools when this one process drives several GPUs, else None (one engine)."""
        if self.pool_devices is not None:
            return [torch.device(d) for d in self.pool_devices]
        n_dev = torch.cuda.device_count()
        if self.n_pools < 2 or n_dev < 2 or azdist.world_size() > 1:
            return None
        devs, device_no = [], 1  # examplegenerator.py:144-150: the first pool goes to cuda:1, the count wraps to cuda:0
        for _ in range(self.n_pools):
            if device_no >= n_dev:
                device_no = 0
            devs.append(torch.device("cuda", device_no))
            device_no += 1
        return devs

    def _play_pools(self, n_games, devices):
        """One engine + evaluator per pool, each on its device, driven from this thread (engine.run_selfplay_pools); the pools'
        packed records are brought to self.device and laid out like the ranks of an all-gather."""
        from .engine import run_selfplay_pools
        n_each = int(n_games / len(devices))
        if n_each < 1:
            raise ValueError("n_games=%d is fewer than the %d pools" % (n_games, len(devices)))
        n_slots = int(self.n_slots or min(n_each, 4096))
        engines, evaluators = [], []
        try:
            for i, dev in enumerate(devices):
                with torch.cuda.device(dev):
                    engines.append(SelfPlayEngine(self.game, n_slots, max_games=n_each, device=dev,
                                                  seed=self.seed + 1000003 * self._generation + 7919 * i, **self._engine_kwargs()))
                    if self.eval_backend == "fused":
                        from .fusednet import FusedNet
                        evaluators.append(FusedNet(self.net, dev, max_boards=n_slots, precision=self.eval_precision))
                    else:
                        evaluators.append(DeviceEvaluator(copy.deepcopy(self.net), dev, dtype=self.eval_dtype))
            progs = run_selfplay_pools(engines, evaluators, n_each, use_graph=self.use_graph)
            self.last_progress = _sum_progress(progs)
            bufs = []
            for e in engines:
                with torch.cuda.device(e.device):
                    b = e.export_device()
                    torch.cuda.current_stream(e.device).synchronize()
                bufs.append(b.to(self.device))
            dims = (engines[0].max_plies, engines[0].max_children)
        finally:
            for e in engines:  # (a pool that raised leaves the others' enqueued batches behind: let every device drain first)
                try:
                    torch.cuda.synchronize(e.device)
                except Exception:
                    pass
                e.close()
            for ev in evaluators:
                if hasattr(ev, "close"):
                    ev.close()
        self._generation += 1
        return torch.cat(bufs), bufs[0].numel(), n_each, len(devices), dims

    def _play_and_gather(self, n_games):
        """This rank's shard of the generation on the HIP engine, then the generation-end exchange on DEVICE buffers.
        -> (gathered uint8 device tensor [world * nbytes], nbytes per rank, games per rank, world, (max_plies, max_children))"""
        pool_devs = self._pool_device_list()
        if pool_devs is not None:
            return self._play_pools(n_games, pool_devs)
        world, rank = azdist.world_size(), azdist.rank()
        if world == 1:  # the reference plays int(n_games / n_pools) games in each of its n_pools pools (examplegenerator.py:149):
            n_local = int(n_games / self.n_pools) * self.n_pools if self.n_pools > 1 else int(n_games)  # same count here
        else:
            n_local = int(n_games / world)
        if n_local < 1:
            raise ValueError("n_games=%d is fewer than the %d ranks / pools" % (n_games, max(world, self.n_pools)))
        if world > 1 or not azdist._single():
            # every handler gets a copy of THE current net (examplegenerator.py:121): the training rank's weights and
            # BatchNorm statistics, whatever this rank was constructed with
            self.net = self.net.to(self.device)
            azdist.broadcast_net(self.net, src=0)
        n_slots = int(self.n_slots or min(n_local, 4096))
        engine = SelfPlayEngine(self.game, n_slots, max_games=n_local, device=self.device,
                                seed=self.seed + 1000003 * self._generation + 7919 * rank, **self._engine_kwargs())
        evaluators = []
        try:
            sizes = [n for _, n in slot_groups(n_slots, self.overlap)] if self.overlap > 1 else [n_slots]
            if self.eval_backend == "fused":
                from .fusednet import FusedNet
                evaluators = [FusedNet(self.net, self.device, max_boards=n, precision=self.eval_precision) for n in sizes]
            else:
                evaluators = [DeviceEvaluator(self.net, self.device, dtype=self.eval_dtype) for _ in sizes]
            self.last_progress = run_selfplay(engine, evaluators if self.overlap > 1 else evaluators[0], n_local,
                                              use_graph=self.use_graph, overlap=self.overlap)
            buf = engine.export_device()
            dims = (engine.max_plies, engine.max_children)
            torch.cuda.current_stream(self.device).synchronize()
        finally:
            engine.close()
            for ev in evaluators:
                if hasattr(ev, "close"):
                    ev.close()
        self._generation += 1
        return azdist.all_gather_device_exports(buf), buf.numel(), n_local, world, dims

    def generate_examples(self, n_games):
        """-> list of games; a game is a list of [info_state_str, board (C+1,H,W) f64, pi list[A], z]
        (examplegenerator.py:164-175, game_utils.py:169,200-204).  With torch.distributed initialised the net is broadcast
        from rank 0, each rank plays int(n_games / world_size) games (remainder dropped like int(n_games / n_pools),
        examplegenerator.py:149), the packed records are all-gathered on the device (RCCL) and every rank returns the
        same gathered list.  The reference-format lists are built from ONE device-to-host copy of the gathered buffer."""
        gathered, nbytes, n_local, world, (mp, mc) = self._play_and_gather(n_games)
        host = gathered.cpu().numpy()
        games = []
        for r in range(world):
            ex = unpack_device_export(host[r * nbytes:(r + 1) * nbytes], n_local, mp, mc)
            games.extend(examples_from_export(self.game, ex))
        return games

    def generate_into(self, replay, n_games):
        """The same generation, delivered straight into a DeviceReplay on this rank's GPU (engine -> all-gather -> replay
        store, records never leave HBM and never become Python lists).  Returns the number of games appended."""
        gathered, nbytes, n_local, world, _ = self._play_and_gather(n_games)
        for r in range(world):
            replay.append_device(gathered[r * nbytes:(r + 1) * nbytes], n_local)
        return n_local * world

    def generate_tests(self, n_games, game_fn, n_playouts_mcts):
        """n_games calls of `game_fn` (test_zero_vs_mcts / test_net_vs_mcts / test_zero_vs_random / test_net_vs_random:
        two games each, the agent once as first and once as second player) against an MCTSBot with n_playouts_mcts
        simulations -> average reward sum(score1 + score2) / (2 n_games) (examplegenerator.py:177-195).  All 2 n_games
        games are slots of one device arena (alphazero_openspiel_amd.arena); with torch.distributed the tests shard over
        the ranks and the mean is all-reduced."""
        from . import arena
        name = getattr(game_fn, "__name__", str(game_fn))
        pairing = {"test_zero_vs_mcts": ("zero", "uct"), "test_net_vs_mcts": ("net", "uct"),
                   "test_zero_vs_random": ("zero", "random"), "test_net_vs_random": ("net", "random")}.get(name)
        if pairing is None and name != "test_zero_vs_zero":
            raise NotImplementedError("generate_tests supports the pairings of game_utils.py:53-145, got %s" % name)
        world, rank = azdist.world_size(), azdist.rank()
        # the reference plays int(n_games / n_pools) tests in each of its n_pools pools and divides the summed scores by
        # 2 * n_games AS REQUESTED (examplegenerator.py:149,189): same count (same rule as _play_and_gather) and same divisor
        if world == 1:
            n_local = int(n_games / self.n_pools) * self.n_pools if self.n_pools > 1 else int(n_games)
        else:
            n_local = int(n_games / world)
        if n_local < 1:
            raise ValueError("n_games=%d is fewer than the %d ranks / pools" % (n_games, max(world, self.n_pools)))
        if world > 1:
            self.net = self.net.to(self.device)
            azdist.broadcast_net(self.net, src=0)
        kw = {k: self.kwargs[k] for k in ("n_playouts", "c_puct", "temperature", "keep_search_tree", "use_puct",
                                          "use_probabilistic_actions", "num_probabilistic_actions") if k in self.kwargs}
        statistics = [None] * n_local  # the pairings against bots return no statistics (game_utils.py:65,83)
        if name == "test_zero_vs_zero" and self.generate_statistics:
            # both search trees after every move (game_utils.py:29-31): the reference's own per-game loop over façade bots,
            # one device search per step - an inspection mode (tournament.py:39-52 plays one test per pairing with it)
            from .game_utils import test_zero_vs_zero
            net1 = self.net.to(self.device).eval()
            net2 = (self.net2 if self.net2 is not None else self.net).to(self.device).eval()
            out = [test_zero_vs_zero(net1, None, self.game_name, policy_fn2=net2, generate_statistics=True,
                                     settings1=self.kwargs.get("settings1", kw), settings2=self.kwargs.get("settings2", kw))
                   for _ in range(n_local)]
            s1, s2 = np.array([o[0] for o in out], dtype=np.float64), np.array([o[1] for o in out], dtype=np.float64)
            statistics = [o[2] for o in out]
        elif name == "test_zero_vs_zero":  # two networks, each with its settings (game_utils.py:120-145; net2 defaults to net)
            if world > 1 and self.net2 is not None:
                self.net2 = self.net2.to(self.device)
                azdist.broadcast_net(self.net2, src=0)
            s1, s2, self.last_progress = arena.play_zero_vs_zero(
                self.net, self.net2 if self.net2 is not None else self.net, self.game_name, n_local,
                settings1=self.kwargs.get("settings1", kw), settings2=self.kwargs.get("settings2", kw), device=self.device,
                seed=self.seed + 1000003 * self._generation + 7919 * rank, eval_backend=self.eval_backend,
                eval_precision=self.eval_precision)
        else:
            s1, s2, self.last_progress = self._play_tests(arena, n_local, pairing, n_playouts_mcts, rank, kw)
        self._generation += 1
        total = torch.tensor([float(s1.sum() + s2.sum()), float(2 * n_local)], dtype=torch.float64)
        total = azdist.all_reduce_sum(total, self.device)
        avg_reward = float(total[0]) / (2 * n_games)  # examplegenerator.py:189: sum(examples) / (2 * n_games)
        if self.generate_statistics:  # examplegenerator.py:192-193 (this rank's tests)
            return avg_reward, statistics
        return avg_reward

    def _play_tests(self, arena, n_local, pairing, n_playouts_mcts, rank, kw):
        return arena.play_tests(
            self.net, self.game_name, n_local, pairing[0], pairing[1], opponent_sims=int(n_playouts_mcts), device=self.device,
            seed=self.seed + 1000003 * self._generation + 7919 * rank, n_slots=self.n_slots, eval_backend=self.eval_backend,
            eval_precision=self.eval_precision, **kw)
