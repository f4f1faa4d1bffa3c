"""TEST INFRASTRUCTURE — ctypes binding of oracle/libaz_oracle.so (the C restatement).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libaz_oracle.so")

GAME_CONNECT_FOUR = 0
GAME_BREAKTHROUGH = 1
BACKUPS = {"on-policy": 0, "soft-Z": 1, "A0C": 2, "off-policy": 3}

POLICY_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double),
                        C.POINTER(C.c_double))


class SelfplayCfg(C.Structure):
    _fields_ = [("game", C.c_int32), ("rows", C.c_int32), ("cols", C.c_int32),
                ("n_playouts", C.c_int32), ("use_dirichlet", C.c_int32), ("use_puct", C.c_int32),
                ("keep_search_tree", C.c_int32), ("backup", C.c_int32),
                ("c_puct", C.c_double), ("dirichlet_ratio", C.c_double), ("temperature", C.c_double),
                ("seed", C.c_uint64), ("max_moves", C.c_int32), ("num_probabilistic_actions", C.c_int32)]


class ArenaCfg(C.Structure):
    _fields_ = [("game", C.c_int32), ("rows", C.c_int32), ("cols", C.c_int32), ("n_playouts", C.c_int32),
                ("keep_search_tree", C.c_int32), ("agent", C.c_int32), ("opponent", C.c_int32), ("opponent_sims", C.c_int32),
                ("c_puct", C.c_double), ("temperature", C.c_double), ("opponent_uct_c", C.c_double), ("seed", C.c_uint64),
                ("game_id", C.c_int32), ("sample_plies", C.c_int32)]


class DuelCfg(C.Structure):
    _fields_ = [("game", C.c_int32), ("rows", C.c_int32), ("cols", C.c_int32), ("game_id", C.c_int32),
                ("n_playouts1", C.c_int32), ("n_playouts2", C.c_int32), ("keep_search_tree", C.c_int32), ("reserved", C.c_int32),
                ("c_puct1", C.c_double), ("c_puct2", C.c_double), ("temperature", C.c_double),
                ("agent1", C.c_int32), ("agent2", C.c_int32)]


ARENA_AGENTS = {"zero": 1, "net": 2}
OPPONENTS = {"random": 1, "uct": 2}


def build(force=False):
    if force or not os.path.isfile(LIB_PATH) or \
            os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(HERE, "az_oracle.c")):
        subprocess.check_call(["make", "-C", HERE, "-s"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(LIB_PATH)
    dp, ip, lp, vp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.c_void_p
    L.orc_state_size.restype = C.c_int
    L.orc_num_actions.restype = C.c_int
    L.orc_state_init.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.orc_is_terminal.argtypes = [vp]
    L.orc_current_player.argtypes = [vp]
    L.orc_player_return.argtypes = [vp, C.c_int]
    L.orc_player_return.restype = C.c_double
    L.orc_legal_actions.argtypes = [vp, ip]
    L.orc_apply_action.argtypes = [vp, C.c_int]
    L.orc_state_to_board.argtypes = [vp, dp]
    L.orc_mcts_new.restype = vp
    L.orc_mcts_new.argtypes = [C.c_int, C.c_double, C.c_int, C.c_int, C.c_double, C.c_int, POLICY_FN, vp]
    L.orc_mcts_free.argtypes = [vp]
    L.orc_mcts_playout.argtypes = [vp, vp]
    L.orc_mcts_expand_root_dirichlet.argtypes = [vp, vp, dp]
    L.orc_mcts_visit_counts.argtypes = [vp, dp]
    L.orc_mcts_search.argtypes = [vp, vp, dp, dp]
    L.orc_mcts_update_root.argtypes = [vp, C.c_int]
    L.orc_mcts_root_stats.argtypes = [vp, lp, dp, ip, lp, dp, dp]
    L.orc_mcts_counters.argtypes = [vp, lp]
    L.orc_np_sum.argtypes = [dp, C.c_int]
    L.orc_np_sum.restype = C.c_double
    L.orc_remove_illegal_actions.argtypes = [dp, C.c_int, ip, C.c_int]
    L.orc_play_game_self.argtypes = [C.POINTER(SelfplayCfg), POLICY_FN, vp, dp, C.c_int, dp, C.c_int,
                                     dp, dp, dp, ip, lp, C.c_int, dp, lp]
    L.orc_play_arena_game.argtypes = [C.POINTER(ArenaCfg), POLICY_FN, vp, ip, C.c_int, dp]
    L.orc_play_duel_game.argtypes = [C.POINTER(DuelCfg), POLICY_FN, POLICY_FN, vp, ip, C.c_int, dp]
    L.orc_opponent_action.argtypes = [vp, C.c_int, C.c_int, C.c_double, C.c_uint64, C.c_int32]
    L.orc_nodes_alive.restype = C.c_int64
    L.orc_nodes_total.restype = C.c_int64
    _lib = L
    return L


def parse_game(name):
    """-> (game id, rows, cols)"""
    name = name.strip()
    if name.startswith("connect_four"):
        return GAME_CONNECT_FOUR, 6, 7
    if name.startswith("breakthrough"):
        rows = cols = 8
        if "(" in name:
            for kv in filter(None, (s.strip() for s in name[name.index("(") + 1:name.rindex(")")].split(","))):
                k, v = kv.split("=")
                if k.strip() == "rows":
                    rows = int(v)
                if k.strip() == "columns":
                    cols = int(v)
        return GAME_BREAKTHROUGH, rows, cols
    raise ValueError(name)


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _lp(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


class State:
    """Thin owner of an orc_state (cell-array game state)."""

    def __init__(self, game_name=None, _raw=None):
        L = lib()
        self._buf = C.create_string_buffer(L.orc_state_size())
        if _raw is not None:
            C.memmove(self._buf, _raw, len(self._buf))
            self.game, self.rows, self.cols = np.frombuffer(self._buf, dtype=np.int32, count=3)
        else:
            self.game, self.rows, self.cols = parse_game(game_name)
            L.orc_state_init(self._buf, self.game, self.rows, self.cols)
        self.num_actions = L.orc_num_actions(int(self.game), int(self.rows), int(self.cols))

    @property
    def ptr(self):
        return C.cast(self._buf, C.c_void_p)

    def clone(self):
        return State(_raw=self._buf)

    def is_terminal(self):
        return bool(lib().orc_is_terminal(self.ptr))

    def current_player(self):
        return lib().orc_current_player(self.ptr)

    def player_return(self, p):
        return lib().orc_player_return(self.ptr, p)

    def legal_actions(self):
        out = np.zeros(192, dtype=np.int32)
        n = lib().orc_legal_actions(self.ptr, _ip(out))
        return out[:n].tolist()

    def apply_action(self, a):
        rc = lib().orc_apply_action(self.ptr, int(a))
        if rc != 0:
            raise ValueError("illegal action %d (rc=%d)" % (a, rc))

    def board(self):
        out = np.zeros(4 * int(self.rows) * int(self.cols), dtype=np.float64)
        lib().orc_state_to_board(self.ptr, _dp(out))
        return out.reshape(4, int(self.rows), int(self.cols))


def wrap_policy(py_fn, num_actions, n_cells4):
    """py_fn(board float64[(C+1)*H*W]) -> (priors[A], value).  Returns a ctypes callback (keep it alive)."""

    def cb(user, state_ptr, board_p, priors_p, value_p):
        board = np.ctypeslib.as_array(board_p, shape=(n_cells4,))
        pri, val = py_fn(board)
        out = np.ctypeslib.as_array(priors_p, shape=(num_actions,))
        out[:] = np.asarray(pri, dtype=np.float64)
        value_p[0] = float(val)

    return POLICY_FN(cb)


class MCTS:
    """orc_mcts wrapper (mcts.py:92-203)."""

    def __init__(self, py_policy, game_name, c_puct=2.5, n_playouts=100, use_dirichlet=True,
                 dirichlet_ratio=0.25, use_puct=True):
        L = lib()
        g, r, c = parse_game(game_name)
        self.A = L.orc_num_actions(g, r, c)
        self._cb = wrap_policy(py_policy, self.A, 4 * r * c)
        self._m = L.orc_mcts_new(self.A, c_puct, n_playouts, int(use_dirichlet), dirichlet_ratio,
                                 int(use_puct), self._cb, None)

    def __del__(self):
        if getattr(self, "_m", None):
            lib().orc_mcts_free(self._m)
            self._m = None

    def playout(self, state):
        lib().orc_mcts_playout(self._m, state.clone().ptr)

    def expand_root_dirichlet(self, state, eta):
        eta = np.ascontiguousarray(eta, dtype=np.float64)
        lib().orc_mcts_expand_root_dirichlet(self._m, state.ptr, _dp(eta))

    def search(self, state, eta=None):
        pi = np.zeros(self.A, dtype=np.float64)
        e = np.ascontiguousarray(eta if eta is not None else [0.0], dtype=np.float64)
        lib().orc_mcts_search(self._m, state.ptr, _dp(e), _dp(pi))
        return pi

    def visit_counts(self):
        pi = np.zeros(self.A, dtype=np.float64)
        lib().orc_mcts_visit_counts(self._m, _dp(pi))
        return pi

    def update_root(self, action):
        lib().orc_mcts_update_root(self._m, int(action))

    def root_stats(self):
        rn = np.zeros(1, dtype=np.int64)
        rq = np.zeros(1, dtype=np.float64)
        acts = np.zeros(192, dtype=np.int32)
        cn = np.zeros(192, dtype=np.int64)
        cq = np.zeros(192, dtype=np.float64)
        cp = np.zeros(192, dtype=np.float64)
        n = lib().orc_mcts_root_stats(self._m, _lp(rn), _dp(rq), _ip(acts), _lp(cn), _dp(cq), _dp(cp))
        return {"N": int(rn[0]), "Q": float(rq[0]), "actions": acts[:n].tolist(), "cN": cn[:n].tolist(),
                "cQ": cq[:n].tolist(), "cP": cp[:n].tolist()}

    def counters(self):
        out = np.zeros(5, dtype=np.int64)
        lib().orc_mcts_counters(self._m, _lp(out))
        return dict(zip(("sims", "evals", "terminal_hits", "sum_depth", "sum_children"), out.tolist()))


def np_sum(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return lib().orc_np_sum(_dp(a), len(a))


def remove_illegal_actions(probs, legal):
    p = np.array(probs, dtype=np.float64)
    la = np.ascontiguousarray(legal, dtype=np.int32)
    lib().orc_remove_illegal_actions(_dp(p), len(p), _ip(la), len(la))
    return p


def max_plies(game, rows, cols):
    return rows * cols if game == GAME_CONNECT_FOUR else 2 * cols * (2 * rows - 5) + 1


def play_game_self(py_policy, game_name, n_playouts=100, c_puct=2.5, temperature=1.0, dirichlet_ratio=0.25,
                   use_dirichlet=True, use_puct=True, keep_search_tree=True, backup="on-policy",
                   etas=None, us=None, seed=0, max_moves=0, num_probabilistic_actions=1000, **_ignored):
    """game_utils.py:148-206 through the C restatement.

    Returns dict(examples=[[key, board(4,H,W) f64, pi list[A], value]], actions, root_cN, ret0, counters)."""
    L = lib()
    g, r, c = parse_game(game_name)
    A = L.orc_num_actions(g, r, c)
    mp = max_plies(g, r, c)
    cfg = SelfplayCfg(g, r, c, n_playouts, int(use_dirichlet), int(use_puct), int(keep_search_tree),
                      BACKUPS[backup], c_puct, dirichlet_ratio, temperature, seed, int(max_moves),
                      int(num_probabilistic_actions) if int(num_probabilistic_actions) > 0 else -1)
    cb = wrap_policy(py_policy, A, 4 * r * c)
    stride = 3 * r * c
    eta_arr = None
    if etas is not None:
        eta_arr = np.zeros((mp, stride), dtype=np.float64)
        for i, e in enumerate(etas):
            eta_arr[i, :len(e)] = e
    us_arr = None
    if us is not None:
        us_arr = np.zeros(mp, dtype=np.float64)
        us_arr[:len(us)] = us
    boards = np.zeros((mp, 4 * r * c), dtype=np.float64)
    pis = np.zeros((mp, A), dtype=np.float64)
    values = np.zeros(mp, dtype=np.float64)
    actions = np.zeros(mp, dtype=np.int32)
    cn = np.zeros((mp, stride), dtype=np.int64)
    ret0 = np.zeros(1, dtype=np.float64)
    counters = np.zeros(5, dtype=np.int64)
    n = L.orc_play_game_self(C.byref(cfg), cb, None,
                             _dp(eta_arr) if eta_arr is not None else None, stride,
                             _dp(us_arr) if us_arr is not None else None, mp,
                             _dp(boards), _dp(pis), _dp(values), _ip(actions), _lp(cn), stride,
                             _dp(ret0), _lp(counters))
    if n < 0:
        raise RuntimeError("oracle play_game_self overflow")
    hist = actions[:n].tolist()
    examples = []
    for i in range(n):
        key = ", ".join(str(a) for a in hist[:i])
        examples.append([key, boards[i].reshape(4, r, c).copy(), pis[i].tolist(), float(values[i])])
    return {"examples": examples, "actions": hist,
            "root_cN": [[int(x) for x in row if x >= 0] for row in cn[:n]],
            "ret0": float(ret0[0]),
            "counters": dict(zip(("sims", "evals", "terminal_hits", "sum_depth", "sum_children"),
                                 counters.tolist()))}


def play_arena_game(py_policy, game_name, game_id, agent="zero", opponent="uct", opponent_sims=0, n_playouts=100, c_puct=2.5,
                    temperature=1.0, keep_search_tree=True, opponent_uct_c=1.0, seed=0, use_probabilistic_actions=False,
                    num_probabilistic_actions=1000):
    """game_utils.play_game between the network-driven agent (side game_id & 1) and an opponent bot, through the C
    restatement.  Returns dict(actions, ret0)."""
    L = lib()
    g, r, c = parse_game(game_name)
    A = L.orc_num_actions(g, r, c)
    if agent == "net":
        n_playouts, keep_search_tree = 1, False
    cfg = ArenaCfg(g, r, c, n_playouts, int(keep_search_tree), ARENA_AGENTS[agent], OPPONENTS[opponent], int(opponent_sims),
                   c_puct, temperature, opponent_uct_c, seed, int(game_id),
                   max(0, int(num_probabilistic_actions)) if use_probabilistic_actions and agent == "zero" else 0)
    cb = wrap_policy(py_policy, A, 4 * r * c)
    mp = max_plies(g, r, c)
    actions = np.zeros(mp, dtype=np.int32)
    ret0 = np.zeros(1, dtype=np.float64)
    n = L.orc_play_arena_game(C.byref(cfg), cb, None, _ip(actions), mp, _dp(ret0))
    if n < 0:
        raise RuntimeError("oracle arena game overflow")
    return {"actions": actions[:n].tolist(), "ret0": float(ret0[0])}


def opponent_action(state, opponent, n_sims, uct_c, seed, game_id):
    return lib().orc_opponent_action(state.ptr, OPPONENTS[opponent], int(n_sims), float(uct_c), int(seed), int(game_id))


def play_duel_game(py_policy1, py_policy2, game_name, game_id, n_playouts1=100, n_playouts2=100, c_puct1=2.5, c_puct2=2.5,
                   temperature=1.0, agent1="zero", agent2="zero"):
    """test_zero_vs_zero's play_game through the C restatement (root noise off): bot 1 plays side game_id & 1."""
    L = lib()
    g, r, c = parse_game(game_name)
    A = L.orc_num_actions(g, r, c)
    cfg = DuelCfg(g, r, c, int(game_id), n_playouts1, n_playouts2, 1, 0, c_puct1, c_puct2, temperature,
                  ARENA_AGENTS[agent1], ARENA_AGENTS[agent2])
    cb1, cb2 = wrap_policy(py_policy1, A, 4 * r * c), wrap_policy(py_policy2, A, 4 * r * c)
    mp = max_plies(g, r, c)
    actions = np.zeros(mp, dtype=np.int32)
    ret0 = np.zeros(1, dtype=np.float64)
    n = L.orc_play_duel_game(C.byref(cfg), cb1, cb2, None, _ip(actions), mp, _dp(ret0))
    if n < 0:
        raise RuntimeError("oracle duel overflow")
    return {"actions": actions[:n].tolist(), "ret0": float(ret0[0])}
