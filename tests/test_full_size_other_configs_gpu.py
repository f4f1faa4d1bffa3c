"""BASELINE.json configs 3 and 5 (per-GPU shard) at FULL size, through size-independent properties
(the connect_four headline has tests/test_full_size_gpu.py):

  C3  breakthrough(6x6), 800 sims/move, 10-block x 50 net, 4096 concurrent games
  C5  breakthrough(8x8), 1600 sims/move, 20-block x 50 net, 2048 concurrent games (one GPU's share)

* rules: every recorded game replays through the host rules; the recorded root children are exactly the legal actions,
  ascending; the game ends where the record says, with the recorded return; z alternates and equals the final return;
* conservation of visits under tree reuse (mcts.py:126-153,192-203): first search exactly S below the root, afterwards
  S + max(N_chosen - 1, 0);
* slot-count invariance: the first games of the run replayed on fewer slots are identical, bit for bit;
* compaction invariance (mcts.py:192-203 keeps the chosen subtree; HERE it is copied into the slot's other pool half
  whenever the free tail cannot hold another search): a run whose pools hold only ~3 searches compacts THOUSANDS of
  times and must produce the same records as the run with the default pools (C3: never compacts; C5: about once per
  dozen moves).
"""
import numpy as np
import pytest
import torch

from alphazero_openspiel_amd import games

pytestmark = pytest.mark.gpu

CONFIGS = {
    "C3": dict(game="breakthrough(rows=6,columns=6)", S=800, blocks=10, G=4096, n_small=512),
    "C5": dict(game="breakthrough(rows=8,columns=8)", S=1600, blocks=20, G=2048, n_small=256),
}


def _play(cfg, n_slots, n_games, seed, precision="f16", **engine_kw):
    from alphazero_openspiel_amd import engine as E, fusednet
    from alphazero_openspiel_amd.network import Net
    g = games.load_game(cfg["game"])
    torch.manual_seed(0)
    net = Net(g.information_state_normalized_vector_shape(), g.num_distinct_actions(), n_blocks=cfg["blocks"], n_filters=50).eval()
    fn = fusednet.FusedNet(net, "cuda:0", max_boards=n_slots, precision=precision)
    eng = E.SelfPlayEngine(cfg["game"], n_slots, n_playouts=cfg["S"], max_games=n_games, seed=seed, device=0, **engine_kw)
    prog = E.run_selfplay(eng, fn, n_games, use_graph=True, check_every=256)
    assert prog["games_done"] == n_games and prog["error_flags"] == 0
    ex = eng.export()
    prog["nodes_per_slot"] = eng.sizes.nodes_per_slot
    eng.close()
    fn.close()
    return ex, prog


def _same_records(ex, ex_b, n):
    assert (ex["game_len"][:n] == ex_b["game_len"][:n]).all() and (ex["game_ret0"][:n] == ex_b["game_ret0"][:n]).all()
    live_ply = np.arange(ex_b["move"].shape[1])[None, :] < ex_b["game_len"][:n, None]
    live_child = live_ply[:, :, None] & (np.arange(ex_b["child_visits"].shape[2])[None, None, :] < ex_b["n_children"][:n, :, None])
    for k in ("move", "n_children", "value"):
        assert (ex[k][:n][live_ply] == ex_b[k][:n][live_ply]).all(), k
    assert (ex["states"][:n][live_ply] == ex_b["states"][:n][live_ply]).all()
    assert (ex["child_visits"][:n][live_child] == ex_b["child_visits"][:n][live_child]).all()
    assert (ex["child_action"][:n][live_child] == ex_b["child_action"][:n][live_child]).all()


# both evaluator precisions: "f32x" is the product default (fp32-grade, the reference's Net.forward precision), "f16" the opt-in
@pytest.fixture(scope="module", params=["C3-f16", "C5-f16", "C3-f32x", "C5-f32x"])
def full_run(request):
    tag, precision = request.param.split("-")
    cfg = dict(CONFIGS[tag], precision=precision)
    ex, prog = _play(cfg, cfg["G"], cfg["G"], seed=77, precision=precision)
    return request.param, cfg, ex, prog


def test_games_obey_the_rules_and_conserve_visits(full_run):
    tag, cfg, ex, prog = full_run
    game = games.load_game(cfg["game"])
    S = cfg["S"]
    for g in range(cfg["G"]):
        n = int(ex["game_len"][g])
        s = game.new_initial_state()
        prev = None
        for i in range(n):
            assert [int(x) for x in ex["states"][g, i]] == list(s.bb)
            legal = s.legal_actions()
            nc = int(ex["n_children"][g, i])
            assert ex["child_action"][g, i, :nc].tolist() == legal
            visits = ex["child_visits"][g, i, :nc].astype(np.int64)
            assert int(visits.sum()) == (S if prev is None else S + max(prev - 1, 0))
            a = int(ex["move"][g, i])
            assert a in legal and visits[legal.index(a)] > 0
            prev = int(visits[legal.index(a)])
            s.apply_action(a)
        assert s.is_terminal()
        ret0 = s.returns()[0]
        assert ret0 in (-1.0, 1.0) and float(ex["game_ret0"][g]) == ret0   # breakthrough has no draws
        assert (ex["value"][g, :n] == np.where(np.arange(n) % 2 == 0, ret0, -ret0)).all()
    assert prog["moves"] == int(ex["game_len"].sum())


def test_records_do_not_depend_on_the_number_of_slots(full_run):
    tag, cfg, ex, prog = full_run
    n = cfg["n_small"]
    # half as many slots as games: every slot is refilled once (C5 at f32x: as many slots as games - two generations in a row at
    # 128 boards take 90 s there, and the refill is covered by the other three)
    n_slots = n if tag == "C5-f32x" else n // 2
    ex_b, prog_b = _play(cfg, n_slots, n, seed=77, precision=cfg["precision"])
    _same_records(ex, ex_b, n)


def test_thousands_of_compactions_leave_the_records_unchanged(full_run):
    tag, cfg, ex, prog = full_run
    if cfg["precision"] != "f16":
        pytest.skip("compaction is independent of the evaluator; run once per config")
    n = cfg["n_small"]
    maxc = games.load_game(cfg["game"]).max_children()
    small = 3 * (cfg["S"] + 1) * maxc + 64   # room for three searches: re-rooting compacts every second or third move
    ex_b, prog_b = _play(cfg, n, n, seed=77, nodes_per_slot=small)
    assert prog_b["compactions"] > 1000 and prog_b["nodes_per_slot"] == small
    assert prog["nodes_per_slot"] > 4 * small
    _same_records(ex, ex_b, n)
