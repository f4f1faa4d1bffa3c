"""Re-rooting with pools that hold only ~3 searches (mcts.py:192-203 keeps the chosen subtree; HERE the kept subtree is copied
into a spare pool by extra workgroups of the tick kernel's launch) under every way a caller can schedule the launches:

* captured graphs of 1, 3 and 16 launches (the copies used to be keyed by a host-side parity that a graph
  froze - with an odd number of launches per graph every replay queued into one list and drained the other);
* the dense-row tail switching right after compactions (a slot that sat a tick out kept a stale row);
* slot groups on their own streams (az_engine_advance_slots compacts inline) following whole-engine ticks that compacted, and
  tree read-backs between ticks;
* BASELINE.json configs[4] at its own shape: breakthrough 8x8, 1600 sims, 20-block net, fp32-grade evaluator, two slot groups
  on two HIP streams, small pools - against the single-stream run with default pools.

In every case the records must equal those of the eager run with default pools, bit for bit: compaction and launch scheduling
change where nodes live and when, never what the search computes.
"""
import numpy as np
import pytest
import torch

from alphazero_openspiel_amd import games

pytestmark = pytest.mark.gpu


def _net(game_name, blocks):
    from alphazero_openspiel_amd.network import Net
    g = games.load_game(game_name)
    torch.manual_seed(0)
    return Net(g.information_state_normalized_vector_shape(), g.num_distinct_actions(), n_blocks=blocks, n_filters=50).eval()


def _small_pool(game_name, S):
    return 3 * (S + 1) * games.load_game(game_name).max_children() + 64


def _run(game_name, n_slots, n_games, S, blocks=2, precision="f16", seed=11, overlap=1, **kw):
    from alphazero_openspiel_amd import engine as E, fusednet
    net = _net(game_name, blocks)
    engine_kw = {k: kw.pop(k) for k in ("nodes_per_slot",) if k in kw}
    eng = E.SelfPlayEngine(game_name, n_slots, n_playouts=S, max_games=n_games, seed=seed, device=0, **engine_kw)
    if overlap > 1:
        ev = [fusednet.FusedNet(net, "cuda:0", max_boards=n, precision=precision) for _, n in E.slot_groups(n_slots, overlap)]
    else:
        ev = fusednet.FusedNet(net, "cuda:0", max_boards=n_slots, precision=precision)
    prog = E.run_selfplay(eng, ev, n_games, overlap=overlap, **kw)
    assert prog["games_done"] == n_games and prog["error_flags"] == 0, prog
    ex = eng.export()
    eng.close()
    for e in (ev if isinstance(ev, list) else [ev]):
        e.close()
    return ex, prog


def _assert_same(a, b, n=None):
    n = len(b["game_len"]) if n is None else n
    assert (a["game_len"][:n] == b["game_len"][:n]).all() and (a["game_ret0"][:n] == b["game_ret0"][:n]).all()
    live_ply = np.arange(b["move"].shape[1])[None, :] < b["game_len"][:n, None]
    live_child = live_ply[:, :, None] & (np.arange(b["child_visits"].shape[2])[None, None, :] < b["n_children"][:n, :, None])
    for k in ("move", "n_children", "value"):
        assert (a[k][:n][live_ply] == b[k][:n][live_ply]).all(), k
    assert (a["states"][:n][live_ply] == b["states"][:n][live_ply]).all()
    assert (a["child_visits"][:n][live_child] == b["child_visits"][:n][live_child]).all()
    assert (a["child_action"][:n][live_child] == b["child_action"][:n][live_child]).all()


GAME, SLOTS, N_GAMES, S = "breakthrough(rows=6,columns=6)", 256, 600, 40


@pytest.fixture(scope="module")
def base():
    return _run(GAME, SLOTS, N_GAMES, S, use_graph=False)


@pytest.mark.parametrize("tpg", [1, 3, 16])
@pytest.mark.parametrize("compact_tail", [True, False])
def test_graphs_of_any_number_of_launches_compact_like_the_eager_run(base, tpg, compact_tail):
    ex0, prog0 = base
    assert prog0["compactions"] == 0
    ex, prog = _run(GAME, SLOTS, N_GAMES, S, nodes_per_slot=_small_pool(GAME, S), use_graph=True, ticks_per_graph=tpg,
                    check_every=48, compact_tail=compact_tail)
    assert prog["compactions"] > 2000, prog
    _assert_same(ex0, ex)
    assert prog["sims"] == prog0["sims"] and prog["moves"] == prog0["moves"]


def test_eager_ticks_with_small_pools_and_the_row_switch_after_every_check(base):
    """check_every=1: the dense-row switch can fall on the tick right after any compaction (512 slots: the smallest engine whose
    tail switches rows, engine._tail_levels)."""
    ex0, _ = base
    ex, prog = _run(GAME, 512, N_GAMES, S, nodes_per_slot=_small_pool(GAME, S), use_graph=False, check_every=1)
    assert prog["compactions"] > 2000 and prog["tail_compactions"] >= 1, prog
    _assert_same(ex0, ex)


def test_slot_groups_and_read_backs_after_whole_engine_compactions():
    """Whole-engine ticks (copies by the launch's extra workgroups) interleaved with slot-group ticks (inline copies) and tree
    read-backs: no entry point ever meets a pool whose copy is still pending."""
    from alphazero_openspiel_amd import engine as E, fusednet
    net = _net(GAME, 2)
    n_slots, n_games = 64, 160
    ex0, _ = _run(GAME, n_slots, n_games, S, use_graph=False)
    eng = E.SelfPlayEngine(GAME, n_slots, n_playouts=S, max_games=n_games, seed=11, device=0, nodes_per_slot=_small_pool(GAME, S))
    fn = fusednet.FusedNet(net, "cuda:0", max_boards=n_slots, precision="f16")
    eng.reset(n_games)
    obs, pri, val = eng.alloc_io()
    groups = E.slot_groups(n_slots, 2)
    for t in range(200000):
        if t % 3 == 2:
            for first, n in groups:
                eng.advance_slots(first, n, pri, val, obs)
        else:
            eng.advance(pri, val, obs)
        fn(obs, pri, val)
        if t % 64 == 0:
            for g in (0, n_slots - 1):
                tree = eng.read_tree(g)
                kids = tree["parent"][1:] == 0                                 # a live tree, not a half-copied pool:
                assert int(tree["N"][0]) >= int(tree["N"][1:][kids].sum())     # the root's visits cover its children's
            if eng.games_done() >= n_games:
                break
    prog = eng.progress()
    assert prog["games_done"] == n_games and prog["error_flags"] == 0 and prog["compactions"] > 300, prog
    ex = eng.export()
    eng.close()
    fn.close()
    _assert_same(ex0, ex)


def test_config5_shape_with_two_streams_and_small_pools():
    """BASELINE.json configs[4]: breakthrough(8x8), 1600 sims/move, 20-block ResNet, "overlapped PV-eval / tree-search HIP
    streams" - two slot groups of 128 on two streams, the product-default (fp32-grade) evaluator, pools of three searches
    (slot groups compact inline) - against one stream with default pools."""
    game, S5, n = "breakthrough(rows=8,columns=8)", 1600, 256
    ex0, prog0 = _run(game, n, n, S5, blocks=20, precision="f32x", seed=77, use_graph=True, check_every=256)
    ex, prog = _run(game, n, n, S5, blocks=20, precision="f32x", seed=77, use_graph=True, overlap=2, check_every=256,
                    nodes_per_slot=_small_pool(game, S5))
    assert prog["compactions"] > 1000, prog
    _assert_same(ex0, ex)
    assert prog["sims"] == prog0["sims"]
