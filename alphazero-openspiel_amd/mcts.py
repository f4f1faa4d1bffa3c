"""`MCTS` / `Node` with the reference's interface (mcts.py:10-223), backed by ONE engine slot on the GPU.

`MCTS(policy_fn, A, **kw).search(state)` runs the S sequential playouts of mcts.py:164-180 in the HIP
kernels (manual_moves engine, G = 1) and returns the normalised root visit counts.  `update_root`
maps to az_engine_update_root, `root` is a read-only Node view rebuilt from az_engine_read_tree.

Randomness: the root Dirichlet vector is drawn HERE with `np.random.dirichlet(0.3 * ones(n_legal))`
from numpy's global stream, exactly where the reference draws it (mcts.py:187), and injected into
the engine — so after `np.random.seed(k)` a façade game consumes the same random numbers as a
reference game.

policy_fn: a bound `Net.predict` / an nn.Module (evaluated on the device) or any python
`policy_fn(state) -> (priors[A], value)` (evaluated on the host per leaf; slow, for parity/debug).
"""
import numpy as np
import torch

from . import _lib
from .engine import DeviceEvaluator, SelfPlayEngine
from .games import Game, State


class Node:
    """Read-only snapshot of a tree node: parent, children {action: Node}, P, Q, N (mcts.py:10-20)."""

    def __init__(self, parent, prior_p, use_puct=True):
        self.parent = parent
        self.children = dict()
        self.P = prior_p
        self.Q = 0
        self.N = 0
        self.use_puct = use_puct

    def is_leaf(self):
        return self.children == {}

    def is_root(self):
        return self.parent is None


def _tree_to_nodes(tree):
    nodes = []
    for i in range(len(tree["N"])):
        par = nodes[tree["parent"][i]] if i else None
        nd = Node(par, float(tree["P"][i]))
        nd.N, nd.Q = int(tree["N"][i]), float(tree["Q"][i])
        if par is not None:
            par.children[int(tree["action"][i])] = nd
        nodes.append(nd)
    if nodes[0].N == 0:
        nodes[0].Q = 0
    return nodes[0]


def _as_module(policy_fn):
    if isinstance(policy_fn, torch.nn.Module):
        return policy_fn
    owner = getattr(policy_fn, "__self__", None)
    if isinstance(owner, torch.nn.Module) and getattr(policy_fn, "__name__", "") == "predict":
        return owner
    return None


class MCTS:
    def __init__(self, policy_fn, num_distinct_actions, c_puct=2.5, n_playouts=100, use_dirichlet=True,
                 dirichlet_ratio=0.25, use_puct=True, **kwargs):
        self.num_distinct_actions = num_distinct_actions
        self.c_puct, self.n_playouts = c_puct, n_playouts
        self.use_dirichlet, self.dirichlet_ratio, self.use_puct = use_dirichlet, dirichlet_ratio, use_puct
        self.policy_fn = policy_fn
        self.device = kwargs.get("device", None)
        self._engine = None
        self._history = None  # action history of the engine's root state
        # use_puct=False only governs trees whose root update_root() created from a LEAF root (mcts.py:199-200; the
        # constructor's root is always PUCT, mcts.py:122).  Set when update_root() met a leaf root the engine does not hold.
        self._leaf_update = False
        self._io = None
        self._evaluator = None
        self.n_evals = 0

    # ------------------------------------------------------------------ engine plumbing
    def _ensure_engine(self, state):
        if self._engine is not None:
            return
        game = state.get_game() if hasattr(state, "get_game") else state._game
        if not isinstance(game, Game):
            game = Game(str(game))
        dev = self.device
        if dev is None:
            mod = _as_module(self.policy_fn)
            dev = next(mod.parameters()).device if mod is not None else torch.device("cuda", torch.cuda.current_device())
        self._game = game
        self._engine = SelfPlayEngine(game, 1, n_playouts=self.n_playouts, c_puct=self.c_puct,
                                      use_dirichlet=self.use_dirichlet, dirichlet_ratio=self.dirichlet_ratio,
                                      keep_search_tree=True, manual_moves=True, rng="injected", device=dev,
                                      use_puct=self.use_puct, max_games=1, max_sims_per_tick=max(32, self.n_playouts))
        self._io = self._engine.alloc_io()
        mod = _as_module(self.policy_fn)
        self._evaluator = DeviceEvaluator(mod, self._engine.device) if mod is not None else None

    def _restart(self, history):
        e = self._engine
        e.set_start_prefix(history)
        e.reset(1)
        self._history = list(history)

    def _sync_root(self, state):
        hist = [int(a) for a in state.history()]
        if self._history is None or hist[:len(self._history)] != self._history or len(hist) > len(self._history) + 2:
            if self._leaf_update and hist:  # replay the last move through the engine's update_root: leaf root -> new root
                self._restart(hist[:-1])    # with the configured select rule
                self._engine.update_root([hist[-1]], keep_subtree=True)
                self._history.append(hist[-1])
            else:
                self._restart(hist)
            self._leaf_update = False
            return
        self._leaf_update = False
        for a in hist[len(self._history):]:
            self.update_root(a)

    def _host_eval(self, obs, pri, val):
        e = self._engine
        info = e.read_slot(0)
        waiting_root = info["phase"] == 3
        st = State(self._game)
        st.bb = list(info["bb"] if waiting_root else info["leaf_bb"])
        ply = info["ply"] if waiting_root else info["leaf_ply"]
        st._hist = list(self._history) + [-1] * (ply - len(self._history))  # path actions are not tracked
        p, v = self.policy_fn(st)
        pri.copy_(torch.tensor(np.asarray(p, dtype=np.float32)).reshape(1, -1))
        val.fill_(float(v))

    def adopt_engine(self, other):
        """Take over another façade's engine slot (AlphaZeroBot with keep_search_tree=False builds a new MCTS per
        step: the device resources are reused, the tree is not)."""
        self._engine, self._io, self._evaluator = other._engine, other._io, other._evaluator
        if other._engine is not None:
            self._game = other._game
        self._history = None  # forces a fresh tree at the next search
        other._engine = None

    # ------------------------------------------------------------------ reference interface
    def search(self, state):
        self._ensure_engine(state)
        self._sync_root(state)
        e = self._engine
        if e.read_slot(0)["phase"] == 5:  # search() again at an unchanged root: the reference runs another n_playouts on
            e.update_root([_lib.ACTION_SEARCH_AGAIN])  # the same tree and re-expands the root with fresh noise (mcts.py:164-190)
        obs, pri, val = self._io
        ply = len(self._history)
        if self.use_dirichlet:  # the draw of mcts.py:187, from numpy's global stream
            n_legal = len(state.legal_actions(state.current_player()))
            eta = np.random.dirichlet(0.3 * np.ones(n_legal))
            e.set_injected_rng([[[0.0]] * ply + [list(eta)]], [[0.0] * (ply + 1)], absolute_ply=True)
        else:
            e.set_injected_rng(None, [[0.0] * (ply + 1)], absolute_ply=True)
        for _ in range(4 * self.n_playouts + 16):
            e.advance(pri, val, obs)
            info = e.read_slot(0)
            if info["phase"] == 5:  # search done
                break
            if info["phase"] in (3, 4):
                self.n_evals += 1
                if self._evaluator is not None:
                    self._evaluator(obs, pri, val)
                else:
                    self._host_eval(obs, pri, val)
        else:
            raise RuntimeError("search did not finish: %r" % (e.progress(),))
        e.progress()  # raises on device faults
        return self.get_normalized_visit_counts()

    def get_normalized_visit_counts(self):
        r = self._engine.read_root(0)
        visits = [0] * self.num_distinct_actions
        for a, n in zip(r["actions"], r["cN"]):
            visits[a] = n
        total = sum(visits)
        return [float(v) / total for v in visits]

    def update_root(self, action):
        if self._engine is None:  # the constructor's root is a leaf: mcts.py:199-200 replaces it
            self._leaf_update = True
            return
        e = self._engine
        info = e.read_slot(0)
        if info["phase"] == 0:  # idle slot (terminal position reached): restart lazily on the next search
            self._history = None
            self._leaf_update = True  # a terminal node is never expanded: the reference replaces that leaf root too
            return
        e.update_root([int(action)], keep_subtree=True)
        self._history.append(int(action))
        if e.read_slot(0)["phase"] == 0:
            self._history = None

    def random_rollout(self, state):
        """A policy_fn substitute (mcts.py:205-223): flat priors of 1 and the outcome of one uniformly random play-out, seen by
        the player to move in `state`.  Host side, numpy's global stream, like the reference's."""
        sim = state.clone()
        me = sim.current_player()
        while not sim.is_terminal():
            sim.apply_action(int(np.random.choice(sim.legal_actions())))
        return np.ones(self.num_distinct_actions), sim.player_return(me)

    @property
    def root(self):
        if self._engine is None:
            return Node(None, 0.0)
        return _tree_to_nodes(self._engine.read_tree(0))
