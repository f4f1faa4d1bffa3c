"""Evaluation arena on the device (reference game_utils.py:16-145, examplegenerator.py:177-195,
train.py:238-270) against the C restatement (oracle.play_arena_game).

* exact: with the same evaluator (the deterministic fake policy through HostPolicyEvaluator) and the same Philox keys,
  every game of the device arena - AlphaZeroBot or NeuralNetBot against the UCT random-rollout bot or the uniform random
  bot, agent first or second - is the oracle's game, move for move;
* strength (what the arena is for): the reference's shipped checkpoints beat the rollout bot and the random bot by the
  margins the reference reports (tournament.py:19-22: AlphaZero@100 playouts wins ">99 %" against MCTS@200 on
  breakthrough 6x6);
* the reference-facing entry points: ExampleGenerator(is_test=True).generate_tests and the test_* pairings.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import binding as orc
from oracle import fakepolicy

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("game_name,agent,opponent,S,sims", [
    ("connect_four", "zero", "uct", 24, 40), ("connect_four", "zero", "random", 16, 0),
    ("connect_four", "net", "uct", 1, 30), ("connect_four", "net", "random", 1, 0),
    ("breakthrough(rows=6,columns=6)", "zero", "uct", 12, 20), ("breakthrough(rows=5,columns=4)", "net", "random", 1, 0),
])
def test_device_arena_games_equal_the_oracle(game_name, agent, opponent, S, sims):
    from alphazero_openspiel_amd import arena, engine as E
    n_games, salt, seed = 12, 6, 2024
    eng = arena.arena_engine(game_name, 5, n_games, agent, opponent, opponent_sims=sims, device=0, seed=seed, n_playouts=S)
    A = eng.A
    ev = E.HostPolicyEvaluator(eng, lambda b: fakepolicy.fake_eval(b, A, salt))
    ret0, prog, ex = arena.run_arena(eng, ev, n_games, use_graph=False, check_every=4)
    eng.close()
    assert prog["games_done"] == n_games and prog["error_flags"] == 0
    for gid in range(n_games):
        want = orc.play_arena_game(lambda b: fakepolicy.fake_eval(b, A, salt), game_name, gid, agent=agent, opponent=opponent,
                                   opponent_sims=sims, n_playouts=S, seed=seed)
        n = int(ex["game_len"][gid])
        assert ex["move"][gid, :n].tolist() == want["actions"], (gid, agent, opponent)
        assert float(ret0[gid]) == want["ret0"]
    s1, s2 = arena.pair_scores(ret0)
    assert len(s1) == len(s2) == n_games // 2


def _ckpt(tag, shape, A):
    from alphazero_openspiel_amd.network import load_npz_checkpoint
    return load_npz_checkpoint(os.path.join(GOLDEN, "checkpoint_%s.npz" % tag), shape, A)


def test_shipped_checkpoints_beat_the_rollout_bot_as_the_reference_reports():
    from alphazero_openspiel_amd import arena
    net = _ckpt("breakthrough6", [3, 6, 6], 432)
    s1, s2, prog = arena.play_tests(net, "breakthrough(rows=6,columns=6)", 64, "zero", "uct", opponent_sims=200, device="cuda:0",
                                    seed=1, n_playouts=100, c_puct=2.5)
    assert prog["error_flags"] == 0 and set(np.unique(np.concatenate([s1, s2]))) <= {-1.0, 1.0}
    assert float((s1.sum() + s2.sum()) / 128) > 0.95         # tournament.py:19-22: ">99 %" of the games (1024 / 1024 measured)
    net4 = _ckpt("connect_four", [3, 6, 7], 7)
    s1, s2, _ = arena.play_tests(net4, "connect_four", 64, "zero", "uct", opponent_sims=200, device="cuda:0", seed=2, n_playouts=100)
    assert float((s1.sum() + s2.sum()) / 128) > 0.75         # (985 wins, 8 draws of 1024 measured)
    s1, s2, _ = arena.play_tests(net4, "connect_four", 128, "net", "random", device="cuda:0", seed=3)
    assert float((s1.sum() + s2.sum()) / 256) > 0.6          # the raw network alone beats random play


def test_generate_tests_and_the_reference_pairings():
    from alphazero_openspiel_amd import game_utils
    from alphazero_openspiel_amd.examplegenerator import ExampleGenerator
    net = _ckpt("connect_four", [3, 6, 7], 7).cuda()
    gen = ExampleGenerator(net, "connect_four", torch.device("cuda:0"), is_test=True, temperature=1.0, dirichlet_ratio=0.25,
                           c_puct=2.5, n_pools=1, n_processes=1, seed=4, n_playouts=40)
    avg = gen.generate_tests(16, game_utils.test_zero_vs_mcts, 50)
    assert isinstance(avg, float) and -1.0 <= avg <= 1.0 and gen.last_progress["games_done"] == 32
    avg_net = gen.generate_tests(16, game_utils.test_net_vs_mcts, 50)
    assert -1.0 <= avg_net <= 1.0
    with pytest.raises(NotImplementedError):
        gen.generate_tests(4, lambda *a, **k: None, 10)
    # n_pools > 1: int(n_games / n_pools) * n_pools tests are played, the sum is divided by 2 * n_games as requested
    # (examplegenerator.py:149,189) - 16 tests over 3 pools = 15 tests = 30 games; a net that always loses would score -30/32
    gen3 = ExampleGenerator(net, "connect_four", torch.device("cuda:0"), is_test=True, n_pools=3, n_processes=1, seed=4)
    avg3 = gen3.generate_tests(16, game_utils.test_net_vs_random, 0)
    assert gen3.last_progress["games_done"] == 30 and abs(avg3 * 32 - round(avg3 * 32)) < 1e-9 and abs(avg3) <= 30 / 32
    out = game_utils.test_zero_vs_mcts(net.predict, 30, "connect_four", n_playouts=20, c_puct=2.5)
    assert len(out) == 3 and out[2] is None and out[0] in (-1.0, 0.0, 1.0) and out[1] in (-1.0, 0.0, 1.0)
    out = game_utils.test_net_vs_random(net.predict, "connect_four")
    assert len(out) == 2 and all(v in (-1.0, 0.0, 1.0) for v in out)


@pytest.mark.parametrize("game_name,S1,S2", [("connect_four", 24, 12), ("breakthrough(rows=6,columns=6)", 10, 16)])
def test_two_engine_duel_equals_the_oracle(game_name, S1, S2):
    """test_zero_vs_zero (game_utils.py:120-145): two AlphaZero agents with their own evaluators and settings, one engine each,
    moves handed over by az_engine_exchange_moves.  Root noise off: the games equal the oracle's two-bot games move for move."""
    from alphazero_openspiel_amd import arena, engine as E
    n_games = 10
    kw = dict(max_games=n_games, device=0, use_dirichlet=False, arena_agent="zero", opponent="external")
    ea = E.SelfPlayEngine(game_name, n_games, n_playouts=S1, c_puct=2.5, seed=1, arena_flip=False, **kw)
    eb = E.SelfPlayEngine(game_name, n_games, n_playouts=S2, c_puct=1.5, seed=2, arena_flip=True, **kw)
    A = ea.A
    eva = E.HostPolicyEvaluator(ea, lambda b: fakepolicy.fake_eval(b, A, 3))
    evb = E.HostPolicyEvaluator(eb, lambda b: fakepolicy.fake_eval(b, A, 8))
    ret0, prog, ex = arena.run_duel(ea, eb, eva, evb, n_games, use_graph=False, check_every=4)
    ea.close()
    eb.close()
    assert prog["error_flags"] == 0
    for gid in range(n_games):
        want = orc.play_duel_game(lambda b: fakepolicy.fake_eval(b, A, 3), lambda b: fakepolicy.fake_eval(b, A, 8), game_name, gid,
                                  n_playouts1=S1, n_playouts2=S2, c_puct1=2.5, c_puct2=1.5)
        n = int(ex["game_len"][gid])
        assert ex["move"][gid, :n].tolist() == want["actions"], gid
        assert float(ret0[gid]) == want["ret0"]


@pytest.mark.parametrize("pair", range(5))
def test_device_duels_equal_the_games_of_the_reference_bots(pair):
    """tests/golden/arena.json (produced by the REAL reference: play_game between its AlphaZeroBot / NeuralNetBot instances,
    both seatings): two device arena engines facing each other reproduce those games move for move."""
    from conftest import load_golden
    from alphazero_openspiel_amd import arena, engine as E
    c0, c1 = load_golden("arena.json")[2 * pair: 2 * pair + 2]
    assert c0["bot1_side"] == 0 and c1["bot1_side"] == 1 and c0["game"] == c1["game"]
    engines = []
    for flip in (0, 1):
        kind = c0["agents"][flip]
        kw = dict(n_playouts=1, keep_search_tree=False) if kind == "net" else dict(n_playouts=c0["n_playouts"][flip], c_puct=c0["c_puct"][flip])
        engines.append(E.SelfPlayEngine(c0["game"], 2, max_games=2, device=0, use_dirichlet=False, arena_agent=kind, opponent="external",
                                        arena_flip=bool(flip), seed=flip, **kw))
    A = engines[0].A
    evs = [E.HostPolicyEvaluator(engines[i], (lambda salt: (lambda b: fakepolicy.fake_eval(b, A, salt)))(c0["salts"][i])) for i in (0, 1)]
    ret0, prog, ex = arena.run_duel(engines[0], engines[1], evs[0], evs[1], 2, use_graph=False, check_every=2)
    for e in engines:
        e.close()
    for gid, want in enumerate((c0, c1)):
        n = int(ex["game_len"][gid])
        assert ex["move"][gid, :n].tolist() == want["actions"], (pair, gid)
        assert float(ret0[gid]) == want["ret0"]


def test_zero_vs_zero_entry_points():
    from alphazero_openspiel_amd import arena, game_utils
    from alphazero_openspiel_amd.examplegenerator import ExampleGenerator
    from alphazero_openspiel_amd.network import Net
    strong = _ckpt("connect_four", [3, 6, 7], 7).cuda()
    torch.manual_seed(0)
    weak = Net([3, 6, 7], 7).cuda()          # untrained
    s1, s2, prog = arena.play_zero_vs_zero(strong, weak, "connect_four", 64, settings1=dict(n_playouts=50), settings2=dict(n_playouts=50),
                                           device="cuda:0", seed=5)
    assert prog["error_flags"] == 0 and float((s1.sum() + s2.sum()) / 128) > 0.5      # the trained network wins
    s1, s2, _ = arena.play_zero_vs_zero(strong, None, "connect_four", 32, settings1=dict(n_playouts=30), settings2=dict(n_playouts=30),
                                        device="cuda:0", seed=6)
    assert abs(float((s1.sum() + s2.sum()) / 64)) < 0.6                                 # against itself: no systematic winner
    gen = ExampleGenerator(strong, "connect_four", torch.device("cuda:0"), is_test=True, net2=weak, seed=7,
                           settings1=dict(n_playouts=30), settings2=dict(n_playouts=30))
    avg = gen.generate_tests(16, game_utils.test_zero_vs_zero, 0)
    assert -1.0 <= avg <= 1.0 and avg > 0.0
    out = game_utils.test_zero_vs_zero(strong.predict, 0, "connect_four", policy_fn2=weak.predict,
                                       settings1=dict(n_playouts=20), settings2=dict(n_playouts=20))
    assert len(out) == 3 and out[2] == {} and all(v in (-1.0, 0.0, 1.0) for v in out[:2])


@pytest.mark.parametrize("game_name,opponent,S,sims,npa", [
    ("connect_four", "random", 20, 0, 1000), ("connect_four", "uct", 16, 30, 3), ("breakthrough(rows=6,columns=6)", "random", 10, 0, 1000),
])
def test_probabilistic_arena_agent_equals_the_oracle(game_name, opponent, S, sims, npa):
    """AlphaZeroBot(use_probabilistic_actions=True[, num_probabilistic_actions=n]) outside self-play (alphazerobot.py:34-36,
    81-86; tournament.py:35-36): the agent samples its moves from the tempered visit distribution for the first n plies.
    Same Philox move stream on both sides: the device games are the oracle's, move for move - and differ from the greedy agent's."""
    from alphazero_openspiel_amd import arena, engine as E
    n_games, salt, seed = 12, 6, 77
    games = {}
    for prob in (True, False):
        eng = arena.arena_engine(game_name, 5, n_games, "zero", opponent, opponent_sims=sims, device=0, seed=seed, n_playouts=S,
                                 use_probabilistic_actions=prob, num_probabilistic_actions=npa)
        A = eng.A
        ev = E.HostPolicyEvaluator(eng, lambda b: fakepolicy.fake_eval(b, A, salt))
        ret0, prog, ex = arena.run_arena(eng, ev, n_games, use_graph=False, check_every=4)
        eng.close()
        assert prog["games_done"] == n_games and prog["error_flags"] == 0
        games[prob] = [ex["move"][g, :int(ex["game_len"][g])].tolist() for g in range(n_games)]
        for gid in range(n_games):
            want = orc.play_arena_game(lambda b: fakepolicy.fake_eval(b, A, salt), game_name, gid, agent="zero", opponent=opponent,
                                       opponent_sims=sims, n_playouts=S, seed=seed, use_probabilistic_actions=prob,
                                       num_probabilistic_actions=npa)
            assert games[prob][gid] == want["actions"], (gid, prob)
            assert float(ret0[gid]) == want["ret0"]
    assert games[True] != games[False]


def test_use_puct_false_changes_nothing_in_arena_games_from_the_initial_position():
    """MCTS(use_puct=False) only governs trees that update_root starts from a LEAF root (mcts.py:122,199-200).  A bot that plays
    from the initial position never meets one (first step: no update_root with < 2 moves played, alphazerobot.py:62-64; later
    steps: the root it re-roots from has been searched) - so, as in the reference (test_mcts.py:44-79 plots four PUCT curves),
    the games are the same with either setting, agent first or second."""
    from alphazero_openspiel_amd import arena, engine as E
    n_games, salt = 10, 4
    moves = {}
    for use_puct in (True, False):
        eng = arena.arena_engine("connect_four", 5, n_games, "zero", "random", device=0, seed=31, n_playouts=24, use_puct=use_puct)
        A = eng.A
        ev = E.HostPolicyEvaluator(eng, lambda b: fakepolicy.fake_eval(b, A, salt))
        ret0, prog, ex = arena.run_arena(eng, ev, n_games, use_graph=False, check_every=4)
        eng.close()
        assert prog["error_flags"] == 0
        moves[use_puct] = [ex["move"][g, :int(ex["game_len"][g])].tolist() for g in range(n_games)]
    assert moves[True] == moves[False]


def test_generate_statistics_returns_both_search_trees_after_every_move():
    """generate_statistics (game_utils.py:16-35,120-145; examplegenerator.py:192-193; tournament.py:39-52): per test
    {"game1": {"player1": [{"root": Node}...], "player2": [...]}, "game2": ...}, keyed by network."""
    from alphazero_openspiel_amd import game_utils
    from alphazero_openspiel_amd.examplegenerator import ExampleGenerator
    from alphazero_openspiel_amd.mcts import Node
    net = _ckpt("connect_four", [3, 6, 7], 7).cuda()
    S1, S2 = 24, 12
    s1, s2, stats = game_utils.test_zero_vs_zero(net.predict, 0, "connect_four", generate_statistics=True,
                                                 settings1=dict(n_playouts=S1, use_probabilistic_actions=True),
                                                 settings2=dict(n_playouts=S2, use_probabilistic_actions=True))
    assert s1 in (-1.0, 0.0, 1.0) and s2 in (-1.0, 0.0, 1.0) and set(stats) == {"game1", "game2"}
    for key, first in (("game1", "player1"), ("game2", "player2")):  # network 1 moves first in game 1, second in game 2
        g = stats[key]
        n = len(g["player1"])
        assert n == len(g["player2"]) and 7 <= n <= 42
        second = "player2" if first == "player1" else "player1"
        sims = {"player1": S1, "player2": S2}
        r0, r1 = g[first][0]["root"], g[second][0]["root"]
        assert isinstance(r0, Node) and sum(c.N for c in r0.children.values()) == sims[first] and r0.N == sims[first]
        assert r1.is_leaf() and r1.N == 0                                # the second player has not searched yet
        r1 = g[second][1]["root"]                                        # ... and has after its first move
        assert sum(c.N for c in r1.children.values()) == sims[second]
        assert g[first][1]["root"].N == g[first][0]["root"].N            # unchanged while the other player moved
        deep = g[first][2]["root"]                                       # second search: the tree kept across two moves
        assert deep.N >= sims[first] and any(not c.is_leaf() for c in deep.children.values())
    gen = ExampleGenerator(net, "connect_four", torch.device("cuda:0"), is_test=True, generate_statistics=True, seed=3,
                           settings1=dict(n_playouts=10), settings2=dict(n_playouts=10))
    avg, st = gen.generate_tests(2, game_utils.test_zero_vs_zero, None)
    assert -1.0 <= avg <= 1.0 and len(st) == 2 and all(set(x) == {"game1", "game2"} for x in st)
    avg, st = gen.generate_tests(3, game_utils.test_zero_vs_mcts, 20)
    assert -1.0 <= avg <= 1.0 and st == [None] * 3


def test_arena_configuration_errors():
    from alphazero_openspiel_amd import engine as E
    with pytest.raises(E.EngineError):
        E.SelfPlayEngine("connect_four", 4, arena_agent="zero")                                  # no opponent
    with pytest.raises(E.EngineError):
        E.SelfPlayEngine("connect_four", 4, arena_agent="zero", opponent="uct", opponent_sims=1)  # MCTSBot needs >= 2 simulations
    eng = E.SelfPlayEngine("connect_four", 4, n_playouts=4)
    eng.reset(4)
    with pytest.raises(E.EngineError):
        eng.opponent_moves()                                                                      # not an arena engine
    eng.close()
