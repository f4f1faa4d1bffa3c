"""Overlapped PV-eval / tree-search streams (BASELINE.json configs[4]; reference analogue: the worker pool running beside
the inference server, examplegenerator.py:106-138): k slot groups of ONE engine tick on k HIP streams.

The grouping is scheduling only - random streams are keyed by game id and the net evaluates each board independently -
so the records must equal the single-stream run's, bit for bit, eager or graph-captured, for any k.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(game_name, n_slots, n_games, S, overlap, use_graph, blocks=2, seed=9):
    from alphazero_openspiel_amd import engine as E, fusednet
    from alphazero_openspiel_amd.games import Game
    from alphazero_openspiel_amd.network import Net
    g = Game(game_name)
    torch.manual_seed(0)
    net = Net(g.information_state_normalized_vector_shape(), g.num_distinct_actions(), n_blocks=blocks, n_filters=50).eval()
    eng = E.SelfPlayEngine(game_name, n_slots, n_playouts=S, max_games=n_games, seed=seed, device=0)
    if overlap > 1:
        ev = [fusednet.FusedNet(net, "cuda:0", max_boards=n, precision="f16") for _, n in E.slot_groups(n_slots, overlap)]
    else:
        ev = fusednet.FusedNet(net, "cuda:0", max_boards=n_slots, precision="f16")
    prog = E.run_selfplay(eng, ev, n_games, use_graph=use_graph, overlap=overlap)
    assert prog["games_done"] == n_games and prog["error_flags"] == 0
    ex = eng.export()
    eng.close()
    for e in (ev if isinstance(ev, list) else [ev]):
        e.close()
    return ex, prog


def _assert_same(a, b):
    assert (a["game_len"] == b["game_len"]).all() and (a["game_ret0"] == b["game_ret0"]).all()
    live_ply = np.arange(a["move"].shape[1])[None, :] < a["game_len"][:, None]
    live_child = live_ply[:, :, None] & (np.arange(a["child_visits"].shape[2])[None, None, :] < a["n_children"][:, :, None])
    for k in ("move", "n_children", "value"):
        assert (a[k][live_ply] == b[k][live_ply]).all(), k
    assert (a["states"][live_ply] == b["states"][live_ply]).all()
    assert (a["child_visits"][live_child] == b["child_visits"][live_child]).all()
    assert (a["child_action"][live_child] == b["child_action"][live_child]).all()


@pytest.mark.parametrize("game_name,S", [("connect_four", 48), ("breakthrough(rows=6,columns=6)", 24)])
def test_overlapped_slot_groups_play_the_same_games_as_one_stream(game_name, S):
    base, prog = _run(game_name, 64, 200, S, overlap=1, use_graph=True)
    for overlap, graph in ((2, True), (2, False), (3, True)):
        ex, p = _run(game_name, 64, 200, S, overlap=overlap, use_graph=graph)
        _assert_same(base, ex)
        assert p["sims"] == prog["sims"] and p["moves"] == prog["moves"]


def test_example_generator_overlap_keyword():
    from alphazero_openspiel_amd.examplegenerator import ExampleGenerator
    from alphazero_openspiel_amd.network import Net
    torch.manual_seed(0)
    net = Net([3, 6, 7], 7, n_blocks=2, n_filters=50)
    kw = dict(n_playouts=16, n_slots=32, seed=4)
    a = ExampleGenerator(net, "connect_four", torch.device("cuda:0"), **kw).generate_examples(48)
    b = ExampleGenerator(net, "connect_four", torch.device("cuda:0"), overlap=2, **kw).generate_examples(48)
    assert [[r[0] for r in g] for g in a] == [[r[0] for r in g] for g in b]
    assert all(x[2] == y[2] and x[3] == y[3] for ga, gb in zip(a, b) for x, y in zip(ga, gb))


def test_advance_slots_rejects_bad_ranges():
    from alphazero_openspiel_amd import engine as E
    eng = E.SelfPlayEngine("connect_four", 16, n_playouts=4, device=0)
    eng.reset(16)
    obs, pri, val = eng.alloc_io()
    with pytest.raises(E.EngineError):
        eng.advance_slots(8, 16, pri, val, obs)
    with pytest.raises(E.EngineError):
        eng.advance_slots(-1, 4, pri, val, obs)
    eng.advance_slots(8, 8, pri, val, obs)
    torch.cuda.synchronize()
    assert eng.read_slot(0)["phase"] == 6 and eng.read_slot(8)["phase"] == 3   # only the second group has ticked
    eng.close()
