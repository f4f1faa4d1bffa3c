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
              "backup")


class ExampleGenerator:
    def __init__(self, net, game_name, device, n_pools=1, n_processes=1, **kwargs):
        if kwargs.get("is_test") or kwargs.get("net2") is not None:
            raise NotImplementedError("the evaluation-arena branches (is_test / net2 / generate_tests, "
                                      "reference examplegenerator.py:88-90,100-103,177-195) are outside the "
                                      "self-play hot path this package replaces")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise EngineError("ExampleGenerator needs a HIP device: self-play runs in HIP kernels, there is no "
                              "CPU fallback (got device=%s)" % (device,))
        self.net = copy.deepcopy(net)  # examplegenerator.py:86: a frozen copy of the current net
        self.game_name = game_name
        self.game = Game(game_name)
        self.n_pools, self.n_processes = n_pools, n_processes  # accepted for signature parity; see module doc
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

    def _play_and_gather(self, n_games):
        """This rank's shard of the generation on the HIP engine, then the generation-end exchange on DEVICE buffers.
        -> (gathered uint8 device tensor [world * nbytes], nbytes per rank, games per rank, world, (max_plies, max_children))"""
        world, rank = azdist.world_size(), azdist.rank()
        n_local = int(n_games / world)
        if n_local < 1:
            raise ValueError("n_games=%d is fewer than the %d ranks" % (n_games, world))
        if world > 1:
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
        raise NotImplementedError("evaluation arenas are outside the self-play hot path (see __init__)")
