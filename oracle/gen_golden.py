"""TEST INFRASTRUCTURE — generates tests/golden/* by RUNNING THE REAL REFERENCE.

Run in the build container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.gen_golden

What is recorded (all produced by the unmodified reference modules imported
through oracle/ref_harness.py, driven by oracle/pygames.py states and
oracle/fakepolicy.py as policy_fn):

  mcts_trace.json     MCTS.search playout-by-playout root statistics
                      (mcts.py:126-190), Dirichlet off and on (recorded eta)
  selfplay.json       play_game_self whole games, all four value targets
                      (game_utils.py:148-206), with the recorded Dirichlet
                      vectors and the uniform draw behind every np.random.choice
  remove_illegal.json alphazerobot.remove_illegal_actions edge cases
  net_forward_*.npz   network.Net.forward of the two shipped checkpoints on
                      fixed boards (fp32, CPU) + the checkpoint tensors re-packed
                      as .npz (data fixture; lets the GPU box run the same net)
  replay.json         Trainer.remove_duplicates over two generations + FIFO trim, the net_step batch gather and one
                      net_step update (train.py:95-130,156-201,226-236)
  arena.json          game_utils.play_game between the reference's AlphaZeroBot (outside self-play) and NeuralNetBot
                      instances with different policy functions / settings, both seatings (game_utils.py:16-35,
                      alphazerobot.py:42-120)
  rules_*.json        random playouts of oracle/pygames.py (NOT reference output:
                      OpenSpiel is absent; rules are "parity unpinned")

RNG capture follows SURVEY.md Appendix B: np.random.choice(n,p) ==
searchsorted(cumsum(p)/cumsum(p)[-1], u, 'right') with one random_sample() u —
asserted here against the real call on every draw.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")

from . import fakepolicy, pygames, ref_harness  # noqa: E402


# ------------------------------------------------------------------ RNG taps
class RngTap:
    """Records the Dirichlet vectors and choice-uniforms the reference draws
    from numpy's global legacy stream, without changing the stream."""

    def __init__(self):
        self.etas = []
        self.us = []
        self._orig_dir = np.random.dirichlet
        self._orig_choice = np.random.choice

    def __enter__(self):
        tap = self

        def dirichlet(alpha, size=None):
            out = tap._orig_dir(alpha, size)
            tap.etas.append([float(x) for x in out])
            return out

        def choice(a, size=None, replace=True, p=None):
            assert size is None and p is not None and isinstance(a, (int, np.integer))
            st = np.random.get_state()
            real = tap._orig_choice(a, p=p)
            after = np.random.get_state()
            np.random.set_state(st)
            u = np.random.random_sample()
            cdf = np.cumsum(np.asarray(p, dtype=np.float64))
            cdf /= cdf[-1]
            idx = int(np.searchsorted(cdf, u, side="right"))
            assert idx == int(real), (idx, real)
            now = np.random.get_state()
            assert now[2] == after[2] and (now[1] == after[1]).all()
            tap.us.append(float(u))
            return real

        np.random.dirichlet = dirichlet
        np.random.choice = choice
        return self

    def __exit__(self, *exc):
        np.random.dirichlet = self._orig_dir
        np.random.choice = self._orig_choice


def root_stats(root):
    kids = sorted(root.children.items())
    return {
        "N": int(root.N), "Q": float(root.Q),
        "actions": [int(a) for a, _ in kids],
        "cN": [int(c.N) for _, c in kids],
        "cQ": [float(c.Q) for _, c in kids],
        "cP": [float(c.P) for _, c in kids],
    }


def sparse(vec):
    return {str(i): float(v) for i, v in enumerate(vec) if v != 0.0}


def board_bits(board):
    return "".join(str(int(x)) for x in np.asarray(board).reshape(-1))


# ------------------------------------------------------------------ fixtures
def gen_mcts_traces(ref):
    cases = []
    specs = [
        # (game, prefix moves, n_playouts, use_dirichlet, c_puct, salt, seed)
        ("connect_four", [], 96, False, 2.5, 1, None),
        ("connect_four", [3, 3, 2, 4, 1], 128, False, 2.5, 2, None),
        ("connect_four", [], 64, True, 2.5, 3, 11),
        ("connect_four", [3, 2, 3, 2, 3, 2], 80, True, 1.0, 4, 12),  # win-in-one for p0: terminal hits
        ("breakthrough(rows=6,columns=6)", [], 96, False, 2.5, 5, None),
        ("breakthrough(rows=6,columns=6)", [], 64, True, 2.5, 6, 13),
        ("breakthrough(rows=8,columns=8)", [], 48, True, 2.5, 7, 14),
    ]
    for game_name, prefix, S, use_dir, c_puct, salt, seed in specs:
        game = pygames.load_game(game_name)
        A = game.num_distinct_actions()
        shape = game.information_state_normalized_vector_shape()
        state = game.new_initial_state()
        for a in prefix:
            state.apply_action(a)
        if game_name.startswith("breakthrough") and not prefix:
            # walk a few plies so that captures are on the board
            rng = np.random.RandomState(salt)
            for _ in range(9):
                la = state.legal_actions()
                state.apply_action(la[rng.randint(len(la))])
            prefix = state.history()
        pf = fakepolicy.make_policy_fn(ref.network.state_to_board, shape, A, salt)
        tree = ref.mcts.MCTS(pf, A, c_puct=c_puct, n_playouts=S, use_dirichlet=use_dir,
                             dirichlet_ratio=0.25)
        trace = []
        eta = None
        if seed is not None:
            np.random.seed(seed)
        with RngTap() as tap:
            if use_dir:
                tree.expand_root_dirichlet(state)
                eta = tap.etas[-1]
            after_expand = root_stats(tree.root)
            for _ in range(S):
                tree.playout(state.clone())
                trace.append(root_stats(tree.root))
            pi = tree.get_normalized_visit_counts()
        cases.append({
            "game": game_name, "prefix": [int(a) for a in prefix], "n_playouts": S,
            "use_dirichlet": use_dir, "c_puct": c_puct, "dirichlet_ratio": 0.25, "salt": salt,
            "eta": eta, "after_root_expand": after_expand,
            "trace_cN": [t["cN"] for t in trace],
            "trace_rootQ": [t["Q"] for t in trace],
            "final": trace[-1], "pi": sparse(pi),
        })
    return cases


def gen_mcts_traces_uct(ref):
    """MCTS(use_puct=False) (mcts.py:80).  The constructor's root is always a PUCT node (mcts.py:122) and children inherit
    their parent's rule (mcts.py:64), so the UCT formula only ever runs in a tree whose root update_root() created from a
    LEAF root (mcts.py:199-200).  Each case: optional update_root(prefix[-1]) on the untouched tree, then `n_searches`
    rounds of [expand_root_dirichlet], S playouts (traced), update_root(most visited move)."""
    cases = []
    specs = [
        # (game, prefix, n_playouts, use_dirichlet, c_puct, salt, seed, leaf_update, n_searches)
        ("connect_four", [3], 96, False, 2.5, 31, None, True, 2),
        ("connect_four", [3, 3, 2, 4, 1], 80, True, 1.0, 32, 21, True, 2),
        ("connect_four", [3, 2, 3, 2, 3, 2], 64, False, 2.5, 33, None, True, 1),  # win-in-one: terminal hits
        ("connect_four", [2], 64, False, 2.5, 34, None, False, 2),  # no update_root before the search: stays PUCT
        ("breakthrough(rows=6,columns=6)", [], 96, False, 2.5, 35, None, True, 2),
        ("breakthrough(rows=8,columns=8)", [], 48, True, 2.5, 36, 22, True, 1),
    ]
    for game_name, prefix, S, use_dir, c_puct, salt, seed, leaf_update, n_searches in specs:
        game = pygames.load_game(game_name)
        A = game.num_distinct_actions()
        shape = game.information_state_normalized_vector_shape()
        state = game.new_initial_state()
        for a in prefix:
            state.apply_action(a)
        if game_name.startswith("breakthrough") and not prefix:
            rng = np.random.RandomState(salt)
            for _ in range(9):
                la = state.legal_actions()
                state.apply_action(la[rng.randint(len(la))])
            prefix = state.history()
        pf = fakepolicy.make_policy_fn(ref.network.state_to_board, shape, A, salt)
        tree = ref.mcts.MCTS(pf, A, c_puct=c_puct, n_playouts=S, use_dirichlet=use_dir, dirichlet_ratio=0.25,
                             use_puct=False)
        if leaf_update:
            tree.update_root(prefix[-1])
        if seed is not None:
            np.random.seed(seed)
        searches = []
        for r in range(n_searches):
            trace, eta = [], None
            with RngTap() as tap:
                if use_dir:
                    tree.expand_root_dirichlet(state)
                    eta = tap.etas[-1]
                after_expand = root_stats(tree.root)
                for _ in range(S):
                    tree.playout(state.clone())
                    trace.append(root_stats(tree.root))
            final = trace[-1]
            move = final["actions"][int(np.argmax(final["cN"]))]
            searches.append({"eta": eta, "after_root_expand": after_expand, "trace_cN": [t["cN"] for t in trace],
                             "trace_rootQ": [t["Q"] for t in trace], "final": final, "move": int(move),
                             "root_use_puct": bool(tree.root.use_puct)})
            state.apply_action(move)
            if state.is_terminal():
                break
            tree.update_root(move)
        cases.append({"game": game_name, "prefix": [int(a) for a in prefix], "n_playouts": S, "use_dirichlet": use_dir,
                      "c_puct": c_puct, "dirichlet_ratio": 0.25, "salt": salt, "leaf_update": leaf_update,
                      "searches": searches})
    return cases


SELFPLAY_NPA_SPECS = [  # num_probabilistic_actions (alphazerobot.py:36,81-86): sample the first n moves, then argmax
    ("connect_four", 40, "on-policy", 1.0, {"num_probabilistic_actions": 6}, 41, 121),
    ("connect_four", 30, "soft-Z", 1.0, {"num_probabilistic_actions": 0}, 42, 122),
    ("breakthrough(rows=6,columns=6)", 30, "on-policy", 1.0, {"num_probabilistic_actions": 9}, 43, 123),
]


def gen_selfplay(ref, specs=None):
    games = []
    specs = specs or [
        # (game, n_playouts, backup, temperature, extra kwargs, salt, seed)
        ("connect_four", 50, "on-policy", 1.0, {}, 21, 101),
        ("connect_four", 25, "on-policy", 1.0, {}, 22, 102),
        ("connect_four", 40, "soft-Z", 1.0, {}, 23, 103),
        ("connect_four", 40, "A0C", 1.0, {}, 24, 104),
        ("connect_four", 40, "off-policy", 1.0, {}, 25, 105),
        ("connect_four", 30, "on-policy", 0.5, {}, 26, 106),
        ("connect_four", 30, "on-policy", 1.0, {"use_dirichlet": False}, 27, 107),
        ("connect_four", 30, "on-policy", 1.0, {"keep_search_tree": False}, 28, 108),
        ("connect_four", 60, "on-policy", 1.0, {"c_puct": 1.25, "dirichlet_ratio": 0.4}, 29, 109),
        ("breakthrough(rows=6,columns=6)", 40, "on-policy", 1.0, {}, 31, 111),
        ("breakthrough(rows=6,columns=6)", 30, "off-policy", 1.0, {}, 32, 112),
        ("breakthrough(rows=6,columns=6)", 30, "A0C", 1.0, {}, 33, 113),
        ("breakthrough(rows=8,columns=8)", 24, "on-policy", 1.0, {}, 34, 114),
    ]
    AZB = ref.alphazerobot.AlphaZeroBot
    orig_step = AZB.step
    for game_name, S, backup, T, extra, salt, seed in specs:
        game = pygames.load_game(game_name)
        A = game.num_distinct_actions()
        shape = game.information_state_normalized_vector_shape()
        pf = fakepolicy.make_policy_fn(ref.network.state_to_board, shape, A, salt)
        per_move = []

        def step(self, state, _pm=per_move):
            policy, action = orig_step(self, state)
            _pm.append({"root": root_stats(self.mcts.root), "action": int(action)})
            return policy, action

        kwargs = dict(n_playouts=S, temperature=T, dirichlet_ratio=0.25, c_puct=2.5,
                      backup=backup, tree_strap=False)
        kwargs.update(extra)
        np.random.seed(seed)
        AZB.step = step
        try:
            with RngTap() as tap:
                examples = ref.game_utils.play_game_self(pf, game_name, **kwargs)
        finally:
            AZB.step = orig_step
        assert len(examples) == len(per_move)
        if kwargs.get("use_dirichlet", True):
            assert len(tap.etas) == len(per_move)
        n_sampled = min(len(per_move), max(0, kwargs.get("num_probabilistic_actions", 1000)))
        assert len(tap.us) == n_sampled
        tap.us.extend([0.0] * (len(per_move) - n_sampled))  # plies past num_probabilistic_actions: argmax, no draw
        games.append({
            "game": game_name, "kwargs": kwargs, "salt": salt, "seed": seed,
            "etas": tap.etas, "us": tap.us,
            "moves": per_move,
            "examples": [{"key": ex[0], "board": board_bits(ex[1]),
                          "pi": sparse(ex[2]), "value": float(ex[3])} for ex in examples],
        })
    return games


def gen_remove_illegal(ref):
    f = ref.alphazerobot.remove_illegal_actions
    cases = []
    for probs, legal in [
        ([0.1, 0.2, 0.3, 0.4, 0.0, 0.0, 0.0], [0, 1, 2, 3]),
        ([0.1, 0.2, 0.3, 0.4, 0.0, 0.0, 0.0], [4, 5]),        # all mass illegal -> uniform
        ([0.0] * 7, [1, 3, 6]),
        ([0.5, 0.0, 0.0, 0.25, 0.0, 0.25, 0.0], [0, 3, 5, 6]),
        ([1e-7, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0 - 1e-7], [0, 1]),  # sum below the 1e-6 threshold
        ([1.0 / 3] * 3 + [0.0] * 4, [0, 1, 2]),
    ]:
        out = f(np.array(probs, dtype=np.float64), list(legal))
        cases.append({"probs": probs, "legal": legal, "out": [float(x) for x in out]})
    return cases


def gen_net_forward(ref):
    import torch

    torch.set_num_threads(1)
    out = {}
    for tag, game_name, fname in [
        ("connect_four", "connect_four", "example_model_connect_four.pth"),
        ("breakthrough6", "breakthrough(rows=6,columns=6)", "example_model_breakthrough(6x6).pth"),
    ]:
        game = pygames.load_game(game_name)
        A = game.num_distinct_actions()
        shape = game.information_state_normalized_vector_shape()
        sd = torch.load(os.path.join(ref_harness.REFERENCE_DIR, "models", fname),
                        map_location="cpu", weights_only=True)
        net = ref.network.Net(shape, A)
        net.load_state_dict(sd)
        net.eval()
        rng = np.random.RandomState(7)
        boards = []
        state = game.new_initial_state()
        while len(boards) < 24:
            if state.is_terminal():
                state = game.new_initial_state()
            boards.append(ref.network.state_to_board(state, shape))
            la = state.legal_actions()
            state.apply_action(la[rng.randint(len(la))])
        x = torch.from_numpy(np.array(boards)).float()
        with torch.no_grad():
            p, v = net.forward(x)
        np.savez_compressed(os.path.join(GOLD, "net_forward_%s.npz" % tag),
                            boards=np.array(boards, dtype=np.uint8),
                            p=p.numpy(), v=v.numpy())
        np.savez_compressed(os.path.join(GOLD, "checkpoint_%s.npz" % tag),
                            **{k: t.numpy() for k, t in sd.items()})
        out[tag] = (tuple(p.shape), tuple(v.shape))
    return out


def gen_rules():
    """Random playouts of oracle/pygames.py — NOT reference output."""
    out = {}
    for tag, game_name, n_games in [("connect_four", "connect_four", 12),
                                    ("breakthrough6", "breakthrough(rows=6,columns=6)", 8),
                                    ("breakthrough8", "breakthrough(rows=8,columns=8)", 4),
                                    ("breakthrough5x4", "breakthrough(rows=5,columns=4)", 4)]:
        game = pygames.load_game(game_name)
        rng = np.random.RandomState(1234)
        games = []
        for _ in range(n_games):
            s = game.new_initial_state()
            plies = []
            while not s.is_terminal():
                la = s.legal_actions()
                vec = s.information_state_as_normalized_vector()
                a = la[rng.randint(len(la))]
                plies.append({"legal": la, "obs": "".join(str(int(x)) for x in vec),
                              "player": s.current_player(), "action": int(a)})
                s.apply_action(a)
            games.append({"plies": plies, "returns": s.returns(),
                          "final_obs": "".join(str(int(x)) for x in s.information_state_as_normalized_vector())})
        out[tag] = {"game": game_name, "games": games}
    return out


def gen_replay(ref):
    """Trainer.remove_duplicates (train.py:156-201) over two generations of reference self-play games with the FIFO
    trim in between (train.py:226-236), the net_step batch gather (train.py:107-120) and one net_step update
    (train.py:95-130) with the shipped connect_four checkpoint."""
    import types

    import torch

    Trainer = ref.train.Trainer
    AZB = ref.alphazerobot.AlphaZeroBot
    orig_step = AZB.step
    game_name, S, salt = "connect_four", 12, 41
    game = pygames.load_game(game_name)
    A = game.num_distinct_actions()
    shape = game.information_state_normalized_vector_shape()
    pf = fakepolicy.make_policy_fn(ref.network.state_to_board, shape, A, salt)
    games, records = [], []
    for k in range(10):
        per_move = []

        def step(self, state, _pm=per_move):
            policy, action = orig_step(self, state)
            _pm.append({"actions": root_stats(self.mcts.root)["actions"], "cN": root_stats(self.mcts.root)["cN"],
                        "action": int(action)})
            return policy, action

        np.random.seed(500 + k)
        AZB.step = step
        try:
            ex = ref.game_utils.play_game_self(pf, game_name, n_playouts=S, temperature=1.0, dirichlet_ratio=0.25,
                                               c_puct=2.5, backup="on-policy")
        finally:
            AZB.step = orig_step
        games.append(ex)
        records.append({"moves": per_move, "z0": float(ex[0][3])})

    def snapshot(flat):
        return [{"key": it[0], "pi": [float(x) for x in it[2]], "z": float(it[3])} for it in flat]

    buffer = list(games[:6])
    out1 = Trainer.remove_duplicates([s for g in buffer for s in g])
    snap1 = snapshot(out1)
    buffer += games[6:]
    n_games_buffer = 8
    while len(buffer) > n_games_buffer:
        del buffer[0]
    flat2 = [s for g in buffer for s in g]
    out2 = Trainer.remove_duplicates(flat2)
    snap2 = snapshot(out2)
    np.random.seed(5)
    ids = np.random.randint(len(out2), size=16)
    batch = {"ids": [int(i) for i in ids],
             "x": [board_bits(out2[i][1]) for i in ids],
             "pi": [[float(np.float32(v)) for v in out2[i][2]] for i in ids],
             "z": [float(np.float32(out2[i][3])) for i in ids]}
    # one net_step with the shipped checkpoint (CPU, fp32), the same 16 samples
    sd = torch.load(os.path.join(ref_harness.REFERENCE_DIR, "models", "example_model_connect_four.pth"),
                    map_location="cpu", weights_only=True)
    net = ref.network.Net(shape, A)
    net.load_state_dict(sd)
    net.train()
    torch.set_num_threads(1)
    fake = types.SimpleNamespace(current_net=net, batch_size=16, device=torch.device("cpu"),
                                 criterion_value=torch.nn.MSELoss(), it=0,
                                 optimizer=torch.optim.Adam(net.parameters(), lr=0.001, weight_decay=0.0001))
    np.random.seed(5)
    loss_p, loss_v = Trainer.net_step(fake, out2)
    after = {k: v.detach().numpy() for k, v in net.state_dict().items()}
    step = {"loss_p": float(loss_p), "loss_v": float(loss_v),
            "fc1_bias_after": [float(x) for x in after["fc1.bias"]],
            "conv_w_sum_after": float(np.abs(after["resblock3.conv1.weight"]).sum())}
    return {"game": game_name, "n_playouts": S, "salt": salt, "records": records, "first_generation": 6,
            "n_games_buffer": n_games_buffer, "dedupe1": snap1, "dedupe2": snap2, "batch": batch, "net_step": step}


def gen_arena(ref):
    """game_utils.play_game (game_utils.py:16-35) between the reference's own bots OUTSIDE self-play: AlphaZeroBot
    (use_dirichlet=False, as the test_*_vs_mcts pairings construct it: alphazerobot.py:42-93 with the two-move re-rooting and
    the greedy move) and NeuralNetBot (alphazerobot.py:96-120), each with its own policy function and settings, each pairing
    played with bot 1 as first and as second player.  The opponents of the reference's test functions themselves (OpenSpiel's
    MCTSBot / random bot) are absent here; these games pin the AGENT side of the arena."""
    AZB, NNB = ref.alphazerobot.AlphaZeroBot, ref.alphazerobot.NeuralNetBot
    out = []
    specs = [("connect_four", ("zero", "zero"), 24, 12, 2.5, 1.5, 3, 8),
             ("connect_four", ("net", "zero"), 1, 16, 2.5, 2.5, 5, 6),
             ("connect_four", ("zero", "net"), 20, 1, 3.0, 2.5, 7, 9),
             ("breakthrough(rows=6,columns=6)", ("zero", "zero"), 10, 16, 2.5, 1.5, 3, 8),
             ("breakthrough(rows=5,columns=4)", ("net", "zero"), 1, 12, 2.5, 2.5, 2, 4)]
    for game_name, kinds, S1, S2, c1, c2, salt1, salt2 in specs:
        game = pygames.load_game(game_name)
        A = game.num_distinct_actions()
        shape = game.information_state_normalized_vector_shape()
        pfs = [fakepolicy.make_policy_fn(ref.network.state_to_board, shape, A, salt) for salt in (salt1, salt2)]
        for first in (0, 1):  # the side bot 1 plays
            def make(kind, player, pf, S, c):
                return AZB(game, player, pf, use_dirichlet=False, n_playouts=S, c_puct=c) if kind == "zero" else NNB(game, player, pf)
            b1 = make(kinds[0], first, pfs[0], S1, c1)
            b2 = make(kinds[1], 1 - first, pfs[1], S2, c2)
            moves = []
            for b in (b1, b2):
                orig = b.step

                def step(state, _orig=orig):
                    policy, action = _orig(state)
                    moves.append(int(action))
                    return policy, action
                b.step = step
            ret0 = ref.game_utils.play_game(game, b1 if first == 0 else b2, b2 if first == 0 else b1)
            out.append({"game": game_name, "agents": list(kinds), "bot1_side": first, "n_playouts": [S1, S2], "c_puct": [c1, c2],
                        "salts": [salt1, salt2], "actions": moves, "ret0": float(ret0)})
    return out


def main():
    os.makedirs(GOLD, exist_ok=True)
    ref = ref_harness.load_reference()

    def dump(name, obj):
        with open(os.path.join(GOLD, name), "w") as f:
            json.dump(obj, f, separators=(",", ":"))
        print("wrote", name, os.path.getsize(os.path.join(GOLD, name)), "bytes")

    dump("mcts_trace.json", gen_mcts_traces(ref))
    dump("mcts_trace_uct.json", gen_mcts_traces_uct(ref))
    dump("selfplay.json", gen_selfplay(ref))
    dump("selfplay_npa.json", gen_selfplay(ref, SELFPLAY_NPA_SPECS))
    dump("remove_illegal.json", gen_remove_illegal(ref))
    for tag, blob in gen_rules().items():
        dump("rules_%s.json" % tag, blob)
    print("net_forward:", gen_net_forward(ref))
    dump("replay.json", gen_replay(ref))
    dump("arena.json", gen_arena(ref))


if __name__ == "__main__":
    sys.dont_write_bytecode = True
    main()
