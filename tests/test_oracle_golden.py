"""The C restatement (oracle/az_oracle.c) against fixtures produced by RUNNING the real reference
(oracle/gen_golden.py).  Bit-exact: visit counts are integers, Q/P/pi/values are IEEE doubles compared
with ==."""
import numpy as np
import pytest

from oracle import binding as orc
from oracle import fakepolicy

from conftest import load_golden


def _policy(A, salt):
    def fn(board):
        pri, val = fakepolicy.fake_eval(board, A, salt)
        return pri.astype(np.float64), float(val)
    return fn


def _state_after(game, prefix):
    s = orc.State(game)
    for a in prefix:
        s.apply_action(a)
    return s


@pytest.mark.parametrize("idx", range(7))
def test_mcts_trace_matches_reference(idx):
    case = load_golden("mcts_trace.json")[idx]
    s = _state_after(case["game"], case["prefix"])
    m = orc.MCTS(_policy(s.num_actions, case["salt"]), case["game"], c_puct=case["c_puct"],
                 n_playouts=case["n_playouts"], use_dirichlet=case["use_dirichlet"],
                 dirichlet_ratio=case["dirichlet_ratio"])
    if case["use_dirichlet"]:
        m.expand_root_dirichlet(s, case["eta"])
    assert m.root_stats() == case["after_root_expand"]
    for k in range(case["n_playouts"]):
        m.playout(s)
        st = m.root_stats()
        assert st["cN"] == case["trace_cN"][k], "playout %d" % k
        assert st["Q"] == case["trace_rootQ"][k], "playout %d" % k
    assert m.root_stats() == case["final"]
    pi = m.visit_counts()
    want = np.zeros(s.num_actions)
    for a, p in case["pi"].items():
        want[int(a)] = p
    assert (pi == want).all()


@pytest.mark.parametrize("idx", range(6))
def test_mcts_trace_use_puct_false_matches_reference(idx):
    """MCTS(use_puct=False): mcts.py:80 runs only in trees update_root started from a leaf root (mcts.py:122,199-200)."""
    case = load_golden("mcts_trace_uct.json")[idx]
    s = _state_after(case["game"], case["prefix"])
    m = orc.MCTS(_policy(s.num_actions, case["salt"]), case["game"], c_puct=case["c_puct"],
                 n_playouts=case["n_playouts"], use_dirichlet=case["use_dirichlet"],
                 dirichlet_ratio=case["dirichlet_ratio"], use_puct=False)
    if case["leaf_update"]:
        m.update_root(case["prefix"][-1])
    for srch in case["searches"]:
        if case["use_dirichlet"]:
            m.expand_root_dirichlet(s, srch["eta"])
        assert m.root_stats() == srch["after_root_expand"]
        for k in range(case["n_playouts"]):
            m.playout(s)
            st = m.root_stats()
            assert st["cN"] == srch["trace_cN"][k], "playout %d" % k
            assert st["Q"] == srch["trace_rootQ"][k], "playout %d" % k
        assert m.root_stats() == srch["final"]
        s.apply_action(srch["move"])
        if not s.is_terminal():
            m.update_root(srch["move"])
    assert any(not x["root_use_puct"] for x in case["searches"]) == case["leaf_update"]


@pytest.mark.parametrize("idx", range(13))
def test_play_game_self_matches_reference(idx):
    _check_play_game_self(load_golden("selfplay.json")[idx])


@pytest.mark.parametrize("idx", range(3))
def test_play_game_self_num_probabilistic_actions_matches_reference(idx):
    """alphazerobot.py:36,81-86: sampled moves for the first n plies, argmax afterwards (n = 6, 0, 9)."""
    _check_play_game_self(load_golden("selfplay_npa.json")[idx])


def _check_play_game_self(g):
    kw = dict(g["kwargs"])
    game, rows, cols = orc.parse_game(g["game"])
    A = orc.lib().orc_num_actions(game, rows, cols)
    out = orc.play_game_self(_policy(A, g["salt"]), g["game"], etas=g["etas"] or None, us=g["us"], **kw)
    assert out["actions"] == [m["action"] for m in g["moves"]]
    assert out["root_cN"] == [m["root"]["cN"] for m in g["moves"]]
    assert len(out["examples"]) == len(g["examples"])
    for ex, want in zip(out["examples"], g["examples"]):
        assert ex[0] == want["key"]
        assert "".join(str(int(x)) for x in ex[1].reshape(-1)) == want["board"]
        pi = np.zeros(A)
        for a, p in want["pi"].items():
            pi[int(a)] = p
        assert ex[2] == pi.tolist()
        assert ex[3] == want["value"]


def test_remove_illegal_actions_matches_reference():
    for case in load_golden("remove_illegal.json"):
        out = orc.remove_illegal_actions(case["probs"], case["legal"])
        assert out.tolist() == case["out"]


@pytest.mark.parametrize("n", [1, 3, 7, 8, 9, 15, 16, 17, 100, 127, 128, 129, 255, 256, 432, 768, 1000])
def test_numpy_pairwise_sum_restatement(n):
    rng = np.random.RandomState(n)
    for _ in range(20):
        a = rng.random_sample(n) * (rng.random_sample(n) < 0.3)
        assert orc.np_sum(a) == float(np.sum(a))


@pytest.mark.parametrize("tag", ["connect_four", "breakthrough6", "breakthrough8", "breakthrough5x4"])
def test_rules_agree_with_python_games(tag):
    """C cell-array rules vs oracle/pygames.py playouts (both restated from the public rules;
    parity unpinned against OpenSpiel — see oracle/az_oracle.c header)."""
    blob = load_golden("rules_%s.json" % tag)
    for game in blob["games"]:
        s = orc.State(blob["game"])
        for ply in game["plies"]:
            assert not s.is_terminal()
            assert s.legal_actions() == ply["legal"]
            assert s.current_player() == ply["player"]
            b = s.board()
            assert "".join(str(int(x)) for x in b[:3].reshape(-1)) == ply["obs"]
            assert (b[3] == ply["player"]).all()
            s.apply_action(ply["action"])
        assert s.is_terminal()
        assert [s.player_return(0), s.player_return(1)] == game["returns"]


def test_no_node_leak():
    L = orc.lib()
    before = L.orc_nodes_alive()
    orc.play_game_self(_policy(7, 5), "connect_four", n_playouts=20, seed=3)
    assert L.orc_nodes_alive() == before


@pytest.mark.parametrize("idx", range(10))
def test_arena_games_between_reference_bots(idx):
    """tests/golden/arena.json: game_utils.play_game between the reference's AlphaZeroBot (outside self-play: two-move
    re-rooting, greedy move) and NeuralNetBot instances, both seatings - reproduced by the oracle's two-bot game."""
    c = load_golden("arena.json")[idx]
    g, r, cc = orc.parse_game(c["game"])
    A = orc.lib().orc_num_actions(g, r, cc)
    p1 = lambda b: fakepolicy.fake_eval(b, A, c["salts"][0])
    p2 = lambda b: fakepolicy.fake_eval(b, A, c["salts"][1])
    got = orc.play_duel_game(p1, p2, c["game"], c["bot1_side"], n_playouts1=c["n_playouts"][0], n_playouts2=c["n_playouts"][1],
                             c_puct1=c["c_puct"][0], c_puct2=c["c_puct"][1], agent1=c["agents"][0], agent2=c["agents"][1])
    assert got["actions"] == c["actions"] and got["ret0"] == c["ret0"]
