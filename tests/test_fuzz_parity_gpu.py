"""Randomised differential test: the HIP engine against the C oracle over the whole option space of play_game_self
(game_utils.py:148-206) - game and board size, playouts, slots vs games (refill), value target, temperature, tree reuse, root
noise, c_puct, dirichlet_ratio, num_probabilistic_actions, use_puct, node-pool size (compaction), chained-playout limits.
Same fake policy, same injected random draws on both sides: every move, visit vector, pi, value target and counter must be equal
(integers and IEEE doubles compared with ==).  Seeds are fixed, so a failure names its configuration."""
import numpy as np
import pytest

from oracle import binding as orc
from oracle import fakepolicy

pytestmark = pytest.mark.gpu

GAMES = ["connect_four", "connect_four", "breakthrough(rows=5,columns=4)", "breakthrough(rows=6,columns=6)",
         "breakthrough(rows=4,columns=5)", "breakthrough(rows=6,columns=3)"]


def _config(seed):
    r = np.random.RandomState(1000 + seed)
    game = GAMES[r.randint(len(GAMES))]
    use_dirichlet = bool(r.rand() < 0.75)
    S = int(r.randint(2 if not use_dirichlet else 1, 40))
    n_games = int(r.randint(1, 6))
    kw = dict(n_playouts=S, use_dirichlet=use_dirichlet, backup=["on-policy", "soft-Z", "A0C", "off-policy"][r.randint(4)],
              temperature=[1.0, 1.0, 0.5, 2.0][r.randint(4)], keep_search_tree=bool(r.rand() < 0.8),
              c_puct=float(r.choice([0.5, 1.0, 2.5, 4.0])), dirichlet_ratio=float(r.choice([0.1, 0.25, 0.5])),
              num_probabilistic_actions=int(r.choice([1000, 1000, 0, 3, 7])), use_puct=bool(r.rand() < 0.8))
    eng_only = dict(max_sims_per_tick=int(r.choice([0, 1, 3])), chain_window_us=int(r.choice([0, -1, 2])))
    return game, n_games, int(r.randint(1, n_games + 1)), int(r.randint(100)), bool(r.rand() < 0.3), kw, eng_only, r


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("AZ_FUZZ_N", "24"))))  # AZ_FUZZ_N=600 for a long hunt
def test_random_configuration_equals_the_oracle(seed):
    from test_engine_parity import _run_games
    from alphazero_openspiel_amd import engine as E
    game, n_games, n_slots, salt, small_pool, kw, eng_only, r = _config(seed)
    gid, rows, cols = orc.parse_game(game)
    A = orc.lib().orc_num_actions(gid, rows, cols)
    mp = orc.max_plies(gid, rows, cols)
    mc = min(64, 6 * cols) if gid else 7
    etas, us, want = [], [], []
    for _ in range(n_games):
        e = [r.dirichlet(0.3 * np.ones(3 * rows * cols)).tolist() for _ in range(mp)]
        u = r.random_sample(mp).tolist()
        etas.append(e)
        us.append(u)
        want.append(orc.play_game_self(lambda b: fakepolicy.fake_eval(b, A, salt), game, etas=e, us=u, **kw))
    eta_rows = [[row[:mc] for row in e] for e in etas]
    if small_pool:  # three searches' worth of nodes: re-rooting has to compact
        try:
            games, ex, prog = _run_games(E, game, n_games, n_slots, salt, eta_rows, us, **kw, **eng_only,
                                         nodes_per_slot=3 * (kw["n_playouts"] + 1) * mc + 80)
        except E.EngineError as err:  # a kept subtree can outgrow any fixed pool: the engine says so (fault), it does not
            assert "POOL_EXHAUSTED" in str(err)  # play on - then the configuration is checked with the default pool
            small_pool = False
    if not small_pool:
        games, ex, prog = _run_games(E, game, n_games, n_slots, salt, eta_rows, us, **kw, **eng_only)
    assert prog["error_flags"] == 0, (seed, game, kw, eng_only)
    for i in range(n_games):
        n = len(want[i]["actions"])
        assert int(ex["game_len"][i]) == n, (seed, game, kw)
        assert ex["move"][i, :n].tolist() == want[i]["actions"], (seed, game, kw)
        assert float(ex["game_ret0"][i]) == want[i]["ret0"]
        for j in range(n):
            nc = int(ex["n_children"][i, j])
            assert ex["child_visits"][i, j, :nc].tolist() == want[i]["root_cN"][j], (seed, i, j)
            assert games[i][j][2] == want[i]["examples"][j][2], (seed, i, j)
            assert games[i][j][3] == want[i]["examples"][j][3], (seed, i, j, kw["backup"])
            assert (games[i][j][1] == want[i]["examples"][j][1]).all()
    for key, okey in (("sims", "sims"), ("evals", "evals"), ("sum_depth", "sum_depth"), ("terminal_hits", "terminal_hits"),
                      ("sum_children", "sum_children")):
        assert prog[key] == sum(w["counters"][okey] for w in want), (seed, key)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("AZ_FUZZ_ARENA_N", "12"))))
def test_random_arena_configuration_equals_the_oracle(seed):
    """The evaluation arena (game_utils.py:16-117) over its options: agent (search / raw net), opponent (random / UCT rollout bot and
    its simulation count), playouts, c_puct, temperature, tree reuse, sampled or greedy agent moves, slots vs games."""
    from alphazero_openspiel_amd import arena, engine as E
    r = np.random.RandomState(5000 + seed)
    game = GAMES[r.randint(len(GAMES))]
    agent = "zero" if r.rand() < 0.7 else "net"
    opponent = "uct" if r.rand() < 0.5 else "random"
    sims = int(r.randint(2, 40)) if opponent == "uct" else 0
    n_games = int(r.randint(2, 9))
    n_slots = int(r.randint(1, n_games + 1))
    salt, eng_seed = int(r.randint(100)), int(r.randint(1 << 30))
    kw = dict(n_playouts=int(r.randint(2, 30)), c_puct=float(r.choice([0.5, 1.0, 2.5, 4.0])),
              temperature=float(r.choice([1.0, 1.0, 0.5, 2.0])), keep_search_tree=bool(r.rand() < 0.8),
              use_probabilistic_actions=bool(r.rand() < 0.4), num_probabilistic_actions=int(r.choice([1000, 2, 5])))
    uct_c = float(r.choice([1.0, 1.0, 0.5, 2.0]))
    eng = arena.arena_engine(game, n_slots, n_games, agent, opponent, opponent_sims=sims, device=0, seed=eng_seed,
                             opponent_uct_c=uct_c, **kw)
    A = eng.A
    ev = E.HostPolicyEvaluator(eng, lambda b: fakepolicy.fake_eval(b, A, salt))
    ret0, prog, ex = arena.run_arena(eng, ev, n_games, use_graph=False, check_every=4)
    eng.close()
    assert prog["games_done"] == n_games and prog["error_flags"] == 0
    okw = dict(kw) if agent == "zero" else {}
    for gid in range(n_games):
        want = orc.play_arena_game(lambda b: fakepolicy.fake_eval(b, A, salt), game, gid, agent=agent, opponent=opponent,
                                   opponent_sims=sims, seed=eng_seed, opponent_uct_c=uct_c, **okw)
        n = int(ex["game_len"][gid])
        assert ex["move"][gid, :n].tolist() == want["actions"], (seed, gid, game, agent, opponent, kw)
        assert float(ret0[gid]) == want["ret0"]


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("AZ_FUZZ_FACADE_N", "10"))))
def test_random_mcts_call_sequence_equals_the_oracle(seed):
    """The MCTS object as a caller may drive it (mcts.py:92-203): update_root before the first search (leaf root -> the tree takes
    MCTS.use_puct, mcts.py:199-200), searches, repeated searches at one root (mcts.py:164-180), update_root along a random line of
    play - façade over a one-slot device engine against the oracle's MCTS, root statistics compared after every call."""
    from alphazero_openspiel_amd import games
    from alphazero_openspiel_amd.mcts import MCTS
    from alphazero_openspiel_amd.network import state_to_board
    r = np.random.RandomState(9000 + seed)
    name = GAMES[r.randint(len(GAMES))]
    game = games.load_game(name)
    A, shape = game.num_distinct_actions(), game.information_state_normalized_vector_shape()
    salt = int(r.randint(100))
    kw = dict(c_puct=float(r.choice([0.5, 1.0, 2.5, 4.0])), n_playouts=int(r.randint(2, 40)), use_dirichlet=False,
              use_puct=bool(r.rand() < 0.5))
    m = MCTS(fakepolicy.make_policy_fn(state_to_board, shape, A, salt), A, **kw)
    o = orc.MCTS(lambda b: fakepolicy.fake_eval(b, A, salt), name, **kw)
    s, so = game.new_initial_state(), orc.State(name)
    for _ in range(int(r.randint(0, 4))):  # a random opening before the tree is touched
        a = int(r.choice(s.legal_actions()))
        s.apply_action(a)
        so.apply_action(a)
    if s.history() and r.rand() < 0.7:  # update_root on the untouched (leaf) root
        m.update_root(s.history()[-1])
        o.update_root(s.history()[-1])
    for step in range(int(r.randint(2, 6))):
        if s.is_terminal():
            break
        for _ in range(1 + int(r.rand() < 0.25)):  # sometimes search twice at the same root
            pi = m.search(s)
            want = o.search(so)
            assert list(pi) == want.tolist(), (seed, step, name, kw)
        got, rs = m.root, o.root_stats()
        assert got.N == rs["N"] and sorted(got.children) == rs["actions"]
        assert [got.children[a].N for a in rs["actions"]] == rs["cN"]
        assert [got.children[a].Q for a in rs["actions"]] == rs["cQ"] and [got.children[a].P for a in rs["actions"]] == rs["cP"]
        a = int(r.choice(s.legal_actions()))
        s.apply_action(a)
        so.apply_action(a)
        if not s.is_terminal():
            m.update_root(a)
            o.update_root(a)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("AZ_FUZZ_NET_N", "10"))))
def test_random_net_shape_matches_torch(seed):
    """The fused tower + head over random architectures the reference's Net admits (network.py:21-64: any depth, <= 56 filters
    here), boards and batch sizes, both precisions, against the PyTorch fp32 forward of the same weights on the CPU.
    Tolerances: f16 mode 4e-3 / 8e-3 (tests/test_fused_net.py), fp32-grade mode 2e-5."""
    import torch
    from test_fused_net import _random_boards
    from alphazero_openspiel_amd import fusednet, games
    from alphazero_openspiel_amd.network import Net
    r = np.random.RandomState(13000 + seed)
    if r.rand() < 0.35:
        name = "connect_four"
    else:
        rows, cols = int(r.randint(4, 9)), int(r.randint(3, 9))
        name = "breakthrough(rows=%d,columns=%d)" % (rows, cols)
    game = games.load_game(name)
    shape, A = game.information_state_normalized_vector_shape(), game.num_distinct_actions()
    blocks, filters = int(r.randint(1, 13)), int(r.choice([8, 16, 24, 32, 40, 48, 50, 50, 56, int(r.randint(8, 57))]))
    n = int(r.choice([1, 3, int(r.randint(1, 64)), int(r.randint(64, 700)), int(r.randint(700, 2600))]))
    precision = "f16" if r.rand() < 0.5 else "f32x"
    torch.manual_seed(int(r.randint(1 << 30)))
    net = Net(shape, A, n_blocks=blocks, n_filters=filters).eval()
    boards = _random_boards(game, n, seed)
    with torch.no_grad():
        p, v = net(torch.from_numpy(boards))
    try:
        fn = fusednet.FusedNet(net, "cuda:0", max_boards=max(n, 16), precision=precision)
    except RuntimeError as err:  # the documented limit of the fp32-grade mode: its LDS budget
        assert precision == "f32x" and filters > 50 and shape[1] * shape[2] > 48, (name, blocks, filters, str(err))
        pytest.skip("f32x: 51-56 filters on a > 48-cell board do not fit (az_net_create says so)")
    pf, vf = fn.forward(torch.from_numpy(boards).cuda())
    torch.cuda.synchronize()
    pf, vf = pf.cpu().numpy(), vf.cpu().numpy()
    fn.close()
    cfg = (name, blocks, filters, n, precision)
    assert np.isfinite(pf).all() and np.isfinite(vf).all(), cfg
    assert np.abs(pf.sum(1) - 1).max() < 1e-5, cfg
    dp, dv = np.abs(pf - p.numpy()).max(), np.abs(vf - v.numpy()[:, 0]).max()
    tol_p, tol_v = (4e-3, 8e-3) if precision == "f16" else (2e-5, 2e-5)
    assert dp <= tol_p and dv <= tol_v, (cfg, dp, dv)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("AZ_FUZZ_TAIL_N", "4"))))
def test_random_generation_with_and_without_the_dense_tail_is_the_same(seed):
    """run_selfplay with the thinned-out tail on dense request rows (az_engine_compact_rows / az_engine_advance_rows) against the
    plain one-row-per-slot run: random slot counts (not multiples of anything), fewer or more games than slots (refill, then the
    tail), playouts, game, fused precision, graph or eager ticks - identical records."""
    import torch
    from test_tail_rows_gpu import _same
    from alphazero_openspiel_amd import engine as E
    from alphazero_openspiel_amd.fusednet import FusedNet
    from alphazero_openspiel_amd.network import Net
    r = np.random.RandomState(17000 + seed)
    name = "connect_four" if r.rand() < 0.6 else "breakthrough(rows=5,columns=4)"
    G = int(r.randint(1024, 2700))
    n_games = int(r.randint(G // 3, int(2.5 * G)))
    S = int(r.randint(4, 14))
    precision = "f16" if r.rand() < 0.5 else "f32x"
    use_graph, tpg = bool(r.rand() < 0.7), int(r.choice([1, 4, 16]))
    torch.manual_seed(seed)
    eng0 = E.SelfPlayEngine(name, 8, n_playouts=2, max_games=8, device=0)
    shape, A = eng0.game.information_state_normalized_vector_shape(), eng0.A
    eng0.close()
    net = Net(shape, A, n_blocks=int(r.randint(1, 4)), n_filters=int(r.choice([16, 32, 50]))).cuda().eval()
    out = {}
    for compact in (False, True):
        eng = E.SelfPlayEngine(name, G, n_playouts=S, max_games=n_games, device=0, seed=1000 + seed)
        fn = FusedNet(net, "cuda:0", max_boards=G, precision=precision)
        prog = E.run_selfplay(eng, fn, n_games, use_graph=use_graph, ticks_per_graph=tpg, compact_tail=compact)
        out[compact] = (eng.export(), prog)
        eng.close()
        fn.close()
        assert prog["error_flags"] == 0 and prog["games_done"] == n_games, (name, G, n_games, S)
    _same(out[False][0], out[True][0])
    assert out[True][1]["tail_compactions"] >= 1, (G, n_games)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("AZ_FUZZ_REPLAY_N", "5"))))
def test_random_generations_through_the_replay_store_equal_the_restatement(seed):
    """The device replay store (train.py:156-201,226-236: FIFO of games, remove_duplicates with averaged pi / z, first-occurrence
    order) fed by random generations - few playouts and sometimes no root noise, so positions repeat a lot; capacities that drop old
    games; all value targets - against oracle/pyreplay.py on the same games in the reference's list format."""
    import torch
    from oracle import pyreplay
    from alphazero_openspiel_amd import engine as E, games, replay
    from alphazero_openspiel_amd.network import Net
    r = np.random.RandomState(21000 + seed)
    name = ["connect_four", "breakthrough(rows=5,columns=4)", "breakthrough(rows=6,columns=6)"][r.randint(3)]
    game = games.load_game(name)
    torch.manual_seed(int(r.randint(1 << 30)))
    net = Net(game.information_state_normalized_vector_shape(), game.num_distinct_actions(), n_blocks=1, n_filters=16)
    ev = E.DeviceEvaluator(net, "cuda:0")
    per_gen = int(r.randint(3, 30))
    cap = int(r.randint(per_gen, 3 * per_gen + 1))
    kw = dict(n_playouts=int(r.randint(2, 9)), use_dirichlet=bool(r.rand() < 0.5),
              backup=["on-policy", "soft-Z", "A0C", "off-policy"][r.randint(4)], temperature=float(r.choice([1.0, 0.5])))
    rep = replay.DeviceReplay(name, max_games=cap, device=0)
    rep.set_capacity(cap)
    buffer = []
    for gen in range(int(r.randint(1, 5))):
        eng = E.SelfPlayEngine(name, int(r.randint(1, per_gen + 1)), max_games=per_gen, seed=int(r.randint(1 << 30)), **kw)
        E.run_selfplay(eng, ev, per_gen)
        if r.rand() < 0.5:
            rep.append_engine(eng)
        else:
            rep.append_device(eng.export_device(), per_gen)
        host_games = E.examples_from_export(game, eng.export())
        eng.close()
        buffer = pyreplay.fifo_append(buffer, host_games, cap)
        want = pyreplay.remove_duplicates([s for g in buffer for s in g])
        assert rep.dedupe() == len(want), (seed, name, kw, gen)
        u = rep.read_unique()
        assert [p.tolist() for p in u["pi"]] == [w[2] for w in want], (seed, gen)
        assert u["z"].tolist() == [w[3] for w in want], (seed, gen)
        boards = games.boards_from_bitboards(game, u["bitboards"], u["ply"])
        assert all((boards[k] == want[k][1]).all() for k in range(len(want)))
    rep.stats()  # raises on a device fault (key collision guard, bad index)
    rep.close()
