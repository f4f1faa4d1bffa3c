"""Python handle on the HIP self-play engine (C ABI: include/az_engine.h) + the tick loop.

PyTorch is plumbing here: it owns the observation / prior / value device tensors, the stream and the
PV-net; the search, game dynamics and rollout loop are the HIP kernels in csrc/az_engine.hip.
"""
import copy
import ctypes as C

import numpy as np
import torch

from . import _lib
from .games import Game, boards_from_bitboards


class EngineError(RuntimeError):
    pass


def _device_index(device):
    if isinstance(device, int):
        return device
    device = torch.device(device)
    if device.type != "cuda":
        raise EngineError("the self-play engine runs on a HIP device only (got %s); there is no CPU path" % device)
    return device.index if device.index is not None else torch.cuda.current_device()


class SelfPlayEngine:
    """G concurrent games of AlphaZero self-play on one GPU.

    Keyword names follow the reference's kwargs (mcts.py:96-101, alphazerobot.py:26,34-38,
    game_utils.py:155): n_playouts, c_puct, use_dirichlet, dirichlet_ratio, temperature,
    keep_search_tree, backup."""

    def __init__(self, game_name, n_slots, n_playouts=100, c_puct=2.5, temperature=1.0, dirichlet_ratio=0.25,
                 use_dirichlet=True, keep_search_tree=True, backup="on-policy", max_games=None, device=0,
                 rng="philox", seed=0, nodes_per_slot=0, max_sims_per_tick=0, chain_window_us=0, manual_moves=False,
                 dirichlet_alpha=0.3, arena_agent=None, opponent=None, opponent_sims=0, opponent_uct_c=1.0, arena_flip=False,
                 use_puct=True, use_probabilistic_actions=False, num_probabilistic_actions=1000, spare_pools=0):
        self.lib = _lib.load()
        self.game = Game(game_name) if isinstance(game_name, str) else game_name
        self.device_index = _device_index(device)
        self.device = torch.device("cuda", self.device_index)
        if backup not in _lib.BACKUPS:
            raise ValueError("backup must be one of %s" % sorted(_lib.BACKUPS))
        cfg = _lib.AzConfig()
        cfg.struct_size = C.sizeof(_lib.AzConfig)
        cfg.game, cfg.rows, cfg.cols = self.game.game_id, self.game.rows, self.game.cols
        cfg.n_slots = int(n_slots)
        cfg.n_playouts = int(n_playouts)
        cfg.use_dirichlet = int(bool(use_dirichlet))
        cfg.keep_search_tree = int(bool(keep_search_tree))
        cfg.backup = _lib.BACKUPS[backup]
        cfg.rng_mode = {"philox": _lib.RNG_PHILOX, "injected": _lib.RNG_INJECTED}[rng]
        cfg.max_sims_per_tick = int(max_sims_per_tick)
        cfg.chain_window_us = int(chain_window_us)
        cfg.device = self.device_index
        cfg.manual_moves = int(bool(manual_moves))
        cfg.nodes_per_slot = int(nodes_per_slot)
        cfg.spare_pools = int(spare_pools)  # 0 = default (n_slots / 16, at least 16)
        cfg.max_games = int(max_games if max_games is not None else n_slots)
        cfg.c_puct = float(c_puct)
        cfg.dirichlet_ratio = float(dirichlet_ratio)
        cfg.dirichlet_alpha = float(dirichlet_alpha)
        cfg.temperature = float(temperature)
        cfg.seed = int(seed) & (2 ** 64 - 1)
        # evaluation arena (alphazero_openspiel_amd.arena): agent "zero" | "net" against opponent "random" | "uct"
        cfg.arena_agent = _lib.ARENA_AGENTS[arena_agent]
        cfg.arena_opponent = _lib.OPPONENTS[opponent]
        cfg.opponent_sims = int(opponent_sims)
        cfg.opponent_uct_c = float(opponent_uct_c)
        cfg.arena_flip = int(bool(arena_flip))
        # MCTS(use_puct=False) (mcts.py:80,199-200): the rule of trees that update_root starts from a leaf root
        cfg.select_rule = _lib.SELECT_PUCT if use_puct else _lib.SELECT_UCT
        # alphazerobot.py:34-36,81-86: an arena "zero" agent samples its moves (self-play always does), for the first n plies
        cfg.arena_probabilistic = int(bool(use_probabilistic_actions) and arena_agent == "zero")
        cfg.num_probabilistic_actions = int(num_probabilistic_actions) if int(num_probabilistic_actions) > 0 else -1
        self.cfg = cfg
        self.backup = backup
        self._h = C.c_void_p()
        rc = self.lib.az_engine_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            raise EngineError("az_engine_create failed (%d): %s" % (rc, self.lib.az_last_error(None).decode()))
        z = _lib.AzSizes()
        self._check(self.lib.az_engine_sizes(self._h, C.byref(z)))
        self.sizes = z
        self.G, self.A = z.n_slots, z.num_actions
        self.obs_shape = (z.obs_planes, z.rows, z.cols)
        self.max_plies, self.max_children = z.max_plies, z.max_children
        self.start_history = []
        self.n_games = 0

    # ------------------------------------------------------------------ plumbing
    def _check(self, rc):
        if rc < 0:
            raise EngineError("engine call failed (%d): %s" % (rc, self.lib.az_last_error(self._h).decode()))
        return rc

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.lib.az_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def alloc_io(self):
        """(obs[G,C+1,H,W], priors[G,A], values[G]) float32 device tensors of the right shape."""
        obs = torch.zeros((self.G,) + self.obs_shape, dtype=torch.float32, device=self.device)
        pri = torch.full((self.G, self.A), 1.0 / self.A, dtype=torch.float32, device=self.device)
        val = torch.zeros((self.G,), dtype=torch.float32, device=self.device)
        return obs, pri, val

    # ------------------------------------------------------------------ C ABI
    def reset(self, n_games, seed=None):
        if seed is None:
            seed = self.cfg.seed
        self._check(self.lib.az_engine_reset(self._h, int(seed) & (2 ** 64 - 1), int(n_games), self._stream()))
        self.n_games = int(n_games)

    def set_start_prefix(self, actions):
        arr = (C.c_int32 * max(1, len(actions)))(*[int(a) for a in actions])
        self._check(self.lib.az_engine_set_start_prefix(self._h, arr, len(actions)))
        self.start_history = [int(a) for a in actions]

    def set_injected_rng(self, etas, us, absolute_ply=False):
        """etas: per game, per ply, the Dirichlet draw (ragged lists ok); us: per game, per ply uniforms.  Lists are
        indexed by plies played since the start position unless absolute_ply (then by the state's ply number)."""
        n = len(us)
        e = np.zeros((n, self.max_plies, self.max_children), dtype=np.float64)
        u = np.zeros((n, self.max_plies), dtype=np.float64)
        off = 0 if absolute_ply else len(self.start_history)
        for g in range(n):
            for i, row in enumerate(etas[g] if etas is not None else []):
                e[g, off + i, :len(row)] = row
            u[g, off:off + len(us[g])] = us[g]
        dp = C.POINTER(C.c_double)
        self._check(self.lib.az_engine_set_injected_rng(self._h, e.ctypes.data_as(dp), u.ctypes.data_as(dp), n))

    def _ptr(self, t, shape):
        if t is None:
            return None
        if t.dtype != torch.float32 or not t.is_contiguous() or t.device != self.device or tuple(t.shape) != shape:
            raise EngineError("expected a contiguous float32 %s tensor on %s, got %s %s on %s"
                              % (shape, self.device, t.dtype, tuple(t.shape), t.device))
        return C.c_void_p(t.data_ptr())

    def advance(self, priors, values, obs):
        """One tick: consume (priors, values) for last tick's requests, search on, write new requests to obs."""
        self._check(self.lib.az_engine_advance(self._h, self._ptr(priors, (self.G, self.A)),
                                               self._ptr(values, (self.G,)),
                                               self._ptr(obs, (self.G,) + self.obs_shape), self._stream()))

    def advance_slots(self, first_slot, n_slots, priors, values, obs):
        """The tick of slots [first_slot, first_slot + n_slots) only, on the current stream; priors / values / obs are the
        whole-engine tensors.  Disjoint slot groups may run concurrently on different streams (see run_selfplay(overlap=))."""
        self._check(self.lib.az_engine_advance_slots(self._h, int(first_slot), int(n_slots),
                                                     self._ptr(priors, (self.G, self.A)), self._ptr(values, (self.G,)),
                                                     self._ptr(obs, (self.G,) + self.obs_shape), self._stream()))

    def compact_rows(self):
        """Tail of a generation (every game handed out): list the slots that still play and switch the request buffers to
        dense rows -> the number of live slots.  Afterwards tick with advance_rows(n_rows >= that number, ...)."""
        n = C.c_int32(0)
        self._check(self.lib.az_engine_compact_rows(self._h, C.byref(n), self._stream()))
        return int(n.value)

    def advance_rows(self, n_rows, priors, values, obs):
        """One tick over the dense list: priors / values / obs are the whole-engine tensors, only their first n_rows rows
        are used (evaluate just those: evaluator(obs[:n_rows], priors[:n_rows], values[:n_rows]))."""
        self._check(self.lib.az_engine_advance_rows(self._h, int(n_rows), self._ptr(priors, (self.G, self.A)),
                                                    self._ptr(values, (self.G,)),
                                                    self._ptr(obs, (self.G,) + self.obs_shape), self._stream()))

    def opponent_moves(self):
        """Arena engines: let the opponent bot choose its move in every slot where it is to move (applied by the next advance)."""
        self._check(self.lib.az_engine_opponent_moves(self._h, self._stream()))

    def exchange_moves(self, other):
        """Two engines facing each other (opponent="external", arena_flip False / True): pass on the moves just played."""
        self._check(self.lib.az_engine_exchange_moves(self._h, other._h, self._stream()))

    def update_root(self, actions, keep_subtree=True):
        arr = (C.c_int32 * self.G)(*[int(a) for a in actions])
        self._check(self.lib.az_engine_update_root(self._h, arr, int(bool(keep_subtree)), self._stream()))

    def progress(self, check=True):
        p = _lib.AzProgress()
        rc = self.lib.az_engine_progress(self._h, C.byref(p), self._stream())
        if check:
            self._check(rc)
        return {name: getattr(p, name) for name, _ in _lib.AzProgress._fields_ if name != "reserved"}

    def games_done(self):
        """Cheap poll (two words) for the tick loop; raises on device faults like progress()."""
        done, flags = C.c_int64(), C.c_uint32()
        rc = self.lib.az_engine_poll(self._h, C.byref(done), C.byref(flags), self._stream())
        if rc < 0:
            self.progress()  # raises with the decoded fault names
            self._check(rc)
        return done.value

    def read_root(self, slot):
        mc = self.max_children
        rn, rq = C.c_int64(), C.c_double()
        acts, cn = (C.c_int32 * mc)(), (C.c_int64 * mc)()
        cq, cp = (C.c_double * mc)(), (C.c_double * mc)()
        n = self._check(self.lib.az_engine_read_root(self._h, slot, C.byref(rn), C.byref(rq), acts, cn, cq, cp))
        return {"N": rn.value, "Q": rq.value, "actions": list(acts[:n]), "cN": list(cn[:n]),
                "cQ": list(cq[:n]), "cP": list(cp[:n])}

    def read_slot(self, slot):
        s = _lib.AzSlotInfo()
        self._check(self.lib.az_engine_read_slot(self._h, slot, C.byref(s)))
        return {"phase": s.phase, "game_id": s.game_id, "ply": s.ply, "sims_done": s.sims_done, "root": s.root,
                "alloc": s.alloc, "bb": [s.bb[0], s.bb[1]], "leaf_bb": [s.leaf_bb[0], s.leaf_bb[1]],
                "leaf_ply": s.leaf_ply, "depth": s.depth}

    def read_tree(self, slot):
        """Whole tree of a slot, breadth-first: dict of numpy arrays parent/action/N/Q/P."""
        n = self._check(self.lib.az_engine_read_tree(self._h, slot, 0, None, None, None, None, None))
        par, act = np.zeros(n, np.int32), np.zeros(n, np.int32)
        N, Q, P = np.zeros(n, np.int64), np.zeros(n, np.float64), np.zeros(n, np.float64)
        ip, lp, dp = C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_double)
        self._check(self.lib.az_engine_read_tree(self._h, slot, n, par.ctypes.data_as(ip), act.ctypes.data_as(ip),
                                                 N.ctypes.data_as(lp), Q.ctypes.data_as(dp), P.ctypes.data_as(dp)))
        return {"parent": par, "action": act, "N": N, "Q": Q, "P": P}

    def export(self):
        """Finished games as numpy arrays (copies)."""
        v = _lib.AzExampleView()
        self._check(self.lib.az_engine_export(self._h, C.byref(v), self._stream()))
        n, mp, mc = v.n_games, v.max_plies, v.max_children

        def arr(ptr, shape, dtype):
            cnt = int(np.prod(shape))
            return np.ctypeslib.as_array(ptr, shape=(cnt,)).view(dtype).reshape(shape).copy()

        return {
            "game_len": arr(v.game_len, (n,), np.int32), "game_ret0": arr(v.game_ret0, (n,), np.float32),
            "states": arr(v.states, (n, mp, 2), np.uint64), "move": arr(v.move, (n, mp), np.uint16),
            "n_children": arr(v.n_children, (n, mp), np.uint8),
            "child_action": arr(v.child_action, (n, mp, mc), np.uint16),
            "child_visits": arr(v.child_visits, (n, mp, mc), np.uint32),
            "value": arr(v.value, (n, mp), np.float64),
            "start_ply": len(self.start_history),
        }


    def export_device(self):
        """Finished games of the generation packed into one uint8 device tensor (layout: include/az_engine.h,
        az_engine_export_device) - the payload of the generation-end all-gather and of DeviceReplay.append_device."""
        nbytes = self._check(self.lib.az_engine_export_device_bytes(self._h))
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        self._check(self.lib.az_engine_export_device(self._h, C.c_void_p(buf.data_ptr()), int(nbytes), self._stream()))
        return buf


def device_export_layout(n_games, max_plies, max_children):
    """(name, dtype, shape, byte offset) of every array in the packed device export + total bytes."""
    n, mp, mc = int(n_games), int(max_plies), int(max_children)
    spec = (("game_len", np.int32, (n,)), ("game_ret0", np.float32, (n,)), ("states", np.uint64, (n, mp, 2)),
            ("move", np.uint16, (n, mp)), ("n_children", np.uint8, (n, mp)), ("child_action", np.uint16, (n, mp, mc)),
            ("child_visits", np.uint32, (n, mp, mc)), ("value", np.float64, (n, mp)))
    out, off = [], 0
    for name, dt, shape in spec:
        out.append((name, dt, shape, off))
        off += (int(np.prod(shape)) * np.dtype(dt).itemsize + 15) & ~15
    return out, off


def unpack_device_export(host_bytes, n_games, max_plies, max_children, start_ply=0):
    """A host copy of a packed device export -> the dict engine.export() returns (arrays are views)."""
    host_bytes = np.ascontiguousarray(host_bytes, dtype=np.uint8)
    layout, total = device_export_layout(n_games, max_plies, max_children)
    if host_bytes.size < total:
        raise ValueError("export buffer holds %d bytes, layout needs %d" % (host_bytes.size, total))
    ex = {"start_ply": int(start_ply)}
    for name, dt, shape, off in layout:
        cnt = int(np.prod(shape)) * np.dtype(dt).itemsize
        ex[name] = host_bytes[off:off + cnt].view(dt).reshape(shape)
    return ex


# ---------------------------------------------------------------------- host logic on engine records
def pi_from_visits(actions, visits, num_actions):
    """MCTS.get_normalized_visit_counts + remove_illegal_actions (mcts.py:155-162, alphazerobot.py:7-18)
    from the recorded root child visit counts, with numpy's own arithmetic -> list[A] of python floats."""
    total = int(np.sum(visits.astype(np.int64)))
    nv = np.zeros(num_actions, dtype=np.float64)
    nv[actions] = visits.astype(np.float64) / float(total)
    s = np.sum(nv)
    if s > 1e-6:
        nv = nv / s
    else:
        nv = np.zeros(num_actions)
        nv[actions] = 1.0 / len(actions)
    return nv.tolist()


def pis_from_visits(actions, visits, n_children, num_actions):
    """pi_from_visits for many recorded plies at once: actions / visits [n, max_children], n_children [n] -> float64 [n, A].
    Row for row the same IEEE operations in the same order (numpy reduces a C-contiguous last axis with the pairwise sum it
    uses for a 1-D array; tests/test_host_logic.py checks the rows against pi_from_visits bit for bit)."""
    n, mc = visits.shape
    live = np.arange(mc)[None, :] < np.asarray(n_children).reshape(-1, 1)
    v = np.where(live, visits, 0).astype(np.int64)
    total = v.sum(axis=1)
    frac = v.astype(np.float64) / np.where(total > 0, total, 1).astype(np.float64)[:, None]
    rows = np.repeat(np.arange(n), mc).reshape(n, mc)
    nv = np.zeros((n, num_actions), dtype=np.float64)
    nv[rows[live], actions.astype(np.int64)[live]] = frac[live]
    s = np.sum(nv, axis=1)
    ok = s > 1e-6
    out = np.zeros_like(nv)
    out[ok] = nv[ok] / s[ok][:, None]
    if not ok.all():  # no visit mass on any legal action: uniform over the children (alphazerobot.py:15-17)
        bad = ~ok & (np.asarray(n_children).reshape(-1) > 0)
        uni = np.zeros_like(nv)
        cnt = np.asarray(n_children).reshape(-1).astype(np.float64)
        sel = live & bad[:, None]
        uni[rows[sel], actions.astype(np.int64)[sel]] = np.repeat(1.0 / cnt[bad], np.asarray(n_children).reshape(-1)[bad])
        out[bad] = uni[bad]
    return out


def examples_from_export(game, ex, start_history=()):
    """Engine records -> the reference's list of games, each a list of `[info_state_str, board (C+1,H,W)
    float64, pi list[A], value]` (game_utils.py:169,200-204; consumed by train.py:109-126,172-198).
    All plies of all games are converted in bulk (one numpy pass for boards, one for pi); what remains per example is
    building its 4-element list."""
    A = game.num_distinct_actions()
    p0 = int(ex["start_ply"])
    lens = np.asarray(ex["game_len"]).astype(np.int64)
    G = len(lens)
    if G == 0:
        return []
    mp = ex["move"].shape[1]
    ply = np.arange(mp)[None, :]
    valid = (ply >= p0) & (ply < p0 + lens[:, None])           # [G, mp], row-major = game by game, ply by ply
    boards = boards_from_bitboards(game, ex["states"][valid], np.broadcast_to(ply, valid.shape)[valid])
    pis = pis_from_visits(ex["child_action"][valid], ex["child_visits"][valid], ex["n_children"][valid], A).tolist()
    values = ex["value"][valid].tolist()
    moves = ex["move"][valid].tolist()
    prefix = ", ".join(str(int(a)) for a in start_history)
    games, k = [], 0
    for n in lens.tolist():
        key, plies = prefix, []
        for i in range(k, k + n):
            plies.append([key, boards[i], pis[i], values[i]])
            key = (key + ", " if key else "") + str(moves[i])
        games.append(plies)
        k += n
    return games


# ---------------------------------------------------------------------- evaluation of the request batch
class DeviceEvaluator:
    """The engine-side replacement of Evaluator + handle_gpu (examplegenerator.py:39-77): one
    `net.forward` over the whole device-resident request batch, no pipes, no host copies.
    dtype: torch.float32 (reference arithmetic) or torch.float16 / torch.bfloat16 autocast."""

    def __init__(self, net, device, dtype=torch.float32, channels_last=False):
        # a private copy: the caller's module (a Trainer's current_net) keeps its device and its train/eval mode
        self.net = copy.deepcopy(net).to(device).eval()
        self.device = torch.device(device)
        self.dtype = dtype
        self.channels_last = channels_last
        if channels_last:
            self.net = self.net.to(memory_format=torch.channels_last)

    @torch.no_grad()
    def __call__(self, obs, priors_out, values_out):
        x = obs.contiguous(memory_format=torch.channels_last) if self.channels_last else obs
        if self.dtype == torch.float32:
            p, v = self.net(x)
        else:
            with torch.autocast("cuda", dtype=self.dtype):
                p, v = self.net(x)
        priors_out.copy_(p)
        values_out.copy_(v.reshape(-1))


class HostPolicyEvaluator:
    """Routes requests through a reference-style `policy_fn(state) -> (priors, value)` on the host
    (Evaluator.evaluate_nn's contract, examplegenerator.py:44-54).  Slow by construction — it exists for
    parity tests and for AlphaZeroBot(policy_fn=<python callable>)."""

    def __init__(self, engine, board_fn):
        """board_fn(board float64 (C+1,H,W)) -> (priors[A], value)"""
        self.engine = engine
        self.board_fn = board_fn

    def __call__(self, obs, priors_out, values_out):
        e = self.engine
        boards = obs.detach().cpu().numpy().astype(np.float64)
        n = boards.shape[0]  # all slots, or the dense rows of a thinned-out generation
        pri = np.empty((n, e.A), dtype=np.float32)
        val = np.empty((n,), dtype=np.float32)
        for g in range(n):
            p, v = self.board_fn(boards[g])
            pri[g] = np.asarray(p, dtype=np.float32)
            val[g] = np.float32(v)
        priors_out.copy_(torch.from_numpy(pri))
        values_out.copy_(torch.from_numpy(val))


def slot_groups(n_slots, k):
    """Split [0, n_slots) into k contiguous groups whose sizes are multiples of 8 (a tower workgroup evaluates 8 boards)."""
    per = -(-n_slots // max(1, int(k)))
    per = -(-per // 8) * 8
    groups, first = [], 0
    while first < n_slots:
        groups.append((first, min(per, n_slots - first)))
        first += per
    return groups


def _tail_levels(n_slots):
    """Row counts worth switching to as a generation thins out.  The fused towers run 256 workgroups per round: the fp32-grade
    one (the default) takes 511 / 276 / 150 us at 4096 / 2048 / 1024 boards with a board per wave, and 120 / 67 us at 512 / 256
    with a board per workgroup (az_tower_x3c_kernel; profiles/r3_tower_vs_boards.txt) - so halve down to 256 rows."""
    return [n for n in (n_slots // 2, n_slots // 4, n_slots // 8, n_slots // 16) if n >= 256]


def run_selfplay(engine, evaluator, n_games, seed=None, check_every=32, max_ticks=None, use_graph=False,
                 on_tick=None, overlap=1, ticks_per_graph=16, compact_tail=True):
    """ExampleGenerator.run_games without processes: tick the engine until n_games are finished.
    Returns the final progress dict.

    overlap = k > 1 (BASELINE.json configs[4]: "overlapped PV-eval / tree-search HIP streams"): the slots are split into k
    groups, each with its own HIP stream (and graph) ticking [az_engine_advance_slots, PV-net forward of the group]; a
    group's tree search and launch gaps then run beside another group's forward.  `evaluator` must then be a list of k
    evaluators, one per group (a FusedNet owns its intermediate buffers).  The games do not depend on the grouping:
    random streams are keyed by game id.

    compact_tail: once every game has been handed to a slot the batch thins out (a generation lasts as long as its longest
    game); when at most half / a quarter of the slots still play, the engine switches to dense request rows
    (az_engine_compact_rows / az_engine_advance_rows) and the network evaluates only those rows.  The games are the same
    (a board's evaluation does not depend on its row)."""
    if overlap > 1:
        return _run_selfplay_overlapped(engine, evaluator, n_games, seed, check_every, max_ticks, use_graph, overlap)
    steps = selfplay_steps(engine, evaluator, n_games, seed, check_every, max_ticks, use_graph, on_tick, ticks_per_graph, compact_tail)
    while True:
        try:
            next(steps)
        except StopIteration as stop:
            return stop.value


def run_selfplay_pools(engines, evaluators, n_games_each, **kw):
    """k engines - one per "pool" of the reference (examplegenerator.py:140-162: n_pools = the amount of GPUs to utilize), each on
    its own device - driven from ONE host thread: every pass enqueues a batch of ticks on each device and only then waits for
    the batch before, so the devices run side by side.  -> list of the final progress dicts."""
    gens = [selfplay_steps(e, ev, n_games_each, **kw) for e, ev in zip(engines, evaluators)]
    out = [None] * len(gens)
    live = list(range(len(gens)))
    while live:
        for i in list(live):
            try:
                with torch.cuda.device(engines[i].device):
                    next(gens[i])
            except StopIteration as stop:
                out[i] = stop.value
                live.remove(i)
    return out


def selfplay_steps(engine, evaluator, n_games, seed=None, check_every=32, max_ticks=None, use_graph=False, on_tick=None,
                   ticks_per_graph=16, compact_tail=True):
    """Generator form of run_selfplay (one stream): yields after every batch of enqueued ticks, BEFORE it waits for them;
    the final progress dict is the generator's return value."""
    engine.reset(n_games, seed)
    obs, pri, val = engine.alloc_io()
    ticks = 0
    graph = None
    n_first = min(int(n_games), engine.G)  # games handed out by the reset
    levels = _tail_levels(engine.G) if (compact_tail and on_tick is None) else []
    rows = None  # None: one row per slot

    def tick():
        if rows is None:
            engine.advance(pri, val, obs)
            evaluator(obs, pri, val)
        else:
            engine.advance_rows(rows, pri, val, obs)
            evaluator(obs[:rows], pri[:rows], val[:rows])

    if use_graph:
        # one tick = [az_advance_kernel, net.forward] replayed as a HIP graph
        torch.cuda.synchronize(engine.device)
        side = torch.cuda.Stream(engine.device)
        side.wait_stream(torch.cuda.current_stream(engine.device))
        with torch.cuda.stream(side):
            for _ in range(2):  # warm-up (MIOpen / workspace allocation must happen outside capture)
                tick()
                ticks += 1
        torch.cuda.current_stream(engine.device).wait_stream(side)
        torch.cuda.synchronize(engine.device)
        # several ticks per captured graph: fewer graph-boundary bubbles on the stream (+2-3 % games/s at 8-16)
        tpg = 1 if on_tick is not None else max(1, min(int(ticks_per_graph), check_every))
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(tpg):
                tick()
    compactions = 0
    while True:
        for _ in range(check_every if graph is None else max(1, check_every // tpg)):
            if graph is not None:
                graph.replay()
                ticks += tpg
            else:
                tick()
                ticks += 1
            if on_tick is not None:
                on_tick(engine, ticks)
        yield ticks  # (a driver of several engines enqueues the others' batches here)
        done = engine.games_done()
        if done >= n_games:
            break
        if max_ticks is not None and ticks >= max_ticks:
            raise EngineError("self-play did not finish within %d ticks: %r" % (max_ticks, engine.progress()))
        # the tail: every game handed out (a finished slot took the next id until they ran out) and few slots still playing
        if levels and min(n_games, n_first + done) >= n_games and n_games - done <= levels[0]:
            live = engine.compact_rows()
            while levels and live <= levels[0]:
                rows = levels.pop(0)
            compactions += 1
            if graph is not None:  # re-capture for the new row count (one eager tick first: new kernel variants set their
                tick()             # attributes on first use, which must not happen inside a capture)
                ticks += 1
                torch.cuda.synchronize(engine.device)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    for _ in range(tpg):
                        tick()
    prog = engine.progress()
    prog["ticks"] = ticks
    prog["tail_compactions"] = compactions
    return prog


class OverlappedTicker:
    """k slot groups of one engine, each ticking on its own HIP stream (optionally as a captured graph)."""

    def __init__(self, engine, evaluators, overlap, use_graph=True, io=None):
        self.engine = engine
        self.groups = slot_groups(engine.G, overlap)
        if not isinstance(evaluators, (list, tuple)) or len(evaluators) != len(self.groups):
            raise ValueError("overlap=%d needs a list of %d evaluators (one per slot group), got %r"
                             % (overlap, len(self.groups), type(evaluators)))
        self.evaluators = list(evaluators)
        self.obs, self.pri, self.val = io if io is not None else engine.alloc_io()
        dev = engine.device
        self.streams = [torch.cuda.Stream(dev) for _ in self.groups]
        self.graphs = []
        torch.cuda.synchronize(dev)
        main = torch.cuda.current_stream(dev)
        for i, st in enumerate(self.streams):
            st.wait_stream(main)
            with torch.cuda.stream(st):
                for _ in range(2):  # warm-up outside capture
                    self._tick(i)
            st.synchronize()
            if use_graph:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=st):
                    self._tick(i)
                self.graphs.append(g)
        self.ticks = 2

    def _tick(self, i):
        first, n = self.groups[i]
        self.engine.advance_slots(first, n, self.pri, self.val, self.obs)
        self.evaluators[i](self.obs[first:first + n], self.pri[first:first + n], self.val[first:first + n])

    def tick(self):
        """One tick of every group (enqueued group after group; the streams run them concurrently)."""
        for i, st in enumerate(self.streams):
            with torch.cuda.stream(st):
                if self.graphs:
                    self.graphs[i].replay()
                else:
                    self._tick(i)
        self.ticks += 1

    def synchronize(self):
        for st in self.streams:
            st.synchronize()


def _run_selfplay_overlapped(engine, evaluators, n_games, seed, check_every, max_ticks, use_graph, overlap):
    engine.reset(n_games, seed)
    torch.cuda.current_stream(engine.device).synchronize()
    tk = OverlappedTicker(engine, evaluators, overlap, use_graph=use_graph)
    while True:
        for _ in range(check_every):
            tk.tick()
        tk.synchronize()
        if engine.games_done() >= n_games:
            break
        if max_ticks is not None and tk.ticks >= max_ticks:
            raise EngineError("self-play did not finish within %d ticks: %r" % (max_ticks, engine.progress()))
    prog = engine.progress()
    prog["ticks"] = tk.ticks
    return prog
