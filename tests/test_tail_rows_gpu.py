"""The thinned-out tail of a generation on dense request rows (az_engine_compact_rows / az_engine_advance_rows).

Once every game has been handed to a slot, finished slots stay idle; the engine then lists the slots that still play and the
network evaluates only that many rows.  Which row a board sits in must not change anything: the records of a generation run
with and without the switch are identical, with the deterministic fake policy (bit-exact by construction) and with the fused
network (a board's evaluation does not depend on its row or on the batch size)."""
import numpy as np
import pytest
import torch

from oracle import fakepolicy

pytestmark = pytest.mark.gpu


def _same(a, b):
    assert (a["game_len"] == b["game_len"]).all() and (a["game_ret0"] == b["game_ret0"]).all()
    played = np.arange(a["move"].shape[1])[None, :] < a["game_len"][:, None]      # entries past a game's end / a ply's child
    assert (a["n_children"][played] == b["n_children"][played]).all()            # count are never written
    kids = played[:, :, None] & (np.arange(a["child_action"].shape[2])[None, None, :] < a["n_children"][:, :, None])
    for k in ("move", "value", "states"):
        assert (a[k][played] == b[k][played]).all(), k
    for k in ("child_action", "child_visits"):
        assert (a[k][kids] == b[k][kids]).all(), k


@pytest.mark.parametrize("game_name,S,G,n_games", [("connect_four", 20, 24, 24), ("connect_four", 16, 16, 40),
                                                    ("breakthrough(rows=5,columns=4)", 12, 12, 12)])
def test_dense_rows_play_the_same_games_manual_switch(game_name, S, G, n_games):
    """Switch by hand at several points of the tail (also twice), fake policy on the host: records equal the plain run's."""
    from alphazero_openspiel_amd import engine as E

    def run(switch_when):
        eng = E.SelfPlayEngine(game_name, G, n_playouts=S, max_games=n_games, device=0, seed=5)
        eng.reset(n_games)
        A = eng.A
        ev = E.HostPolicyEvaluator(eng, lambda b: fakepolicy.fake_eval(b, A, 9))
        obs, pri, val = eng.alloc_io()
        rows, switches = None, list(switch_when)
        for _ in range(200000):
            if rows is None:
                eng.advance(pri, val, obs)
                ev(obs, pri, val)
            else:
                eng.advance_rows(rows, pri, val, obs)
                ev(obs[:rows], pri[:rows], val[:rows])
            done = eng.games_done()
            if done >= n_games:
                break
            if switches and min(n_games, G + done) >= n_games and n_games - done <= switches[0]:
                switches.pop(0)
                live = eng.compact_rows()
                assert live == n_games - done
                rows = live + 3 if live + 3 <= G else live  # any row count >= the live slots will do
        else:
            pytest.fail("games did not finish")
        prog = eng.progress()
        ex = eng.export()
        eng.close()
        assert prog["error_flags"] == 0 and prog["games_done"] == n_games
        return ex

    plain = run([])
    _same(plain, run([G - 2]))
    _same(plain, run([G // 2, G // 4, 2]))


def test_compact_rows_is_refused_while_games_are_handed_out_and_on_arena_engines():
    from alphazero_openspiel_amd import engine as E
    eng = E.SelfPlayEngine("connect_four", 8, n_playouts=8, max_games=64, device=0, seed=1)
    eng.reset(64)
    obs, pri, val = eng.alloc_io()
    with pytest.raises(E.EngineError):
        eng.compact_rows()                      # before the first tick
    eng.advance(pri, val, obs)
    with pytest.raises(E.EngineError):
        eng.compact_rows()                      # 56 games still to hand out
    with pytest.raises(E.EngineError):
        eng.advance_rows(8, pri, val, obs)      # not switched
    eng.close()
    eng = E.SelfPlayEngine("connect_four", 8, n_playouts=8, max_games=8, device=0, arena_agent="zero", opponent="random",
                           use_dirichlet=False)
    eng.reset(8)
    obs, pri, val = eng.alloc_io()
    eng.advance(pri, val, obs)
    with pytest.raises(E.EngineError):
        eng.compact_rows()
    eng.close()
    eng = E.SelfPlayEngine("connect_four", 8, n_playouts=8, max_games=8, device=0, seed=1)
    eng.reset(8)
    obs, pri, val = eng.alloc_io()
    eng.advance(pri, val, obs)
    assert eng.compact_rows() == 8
    with pytest.raises(E.EngineError):
        eng.advance(pri, val, obs)              # the requests live in dense rows now
    with pytest.raises(E.EngineError):
        eng.advance_rows(4, pri, val, obs)      # fewer rows than live slots
    eng.reset(8)                                # back to one row per slot
    eng.advance(pri, val, obs)
    eng.close()


@pytest.mark.parametrize("precision", ["f16", "f32x"])
def test_run_selfplay_tail_compaction_with_the_fused_net(precision):
    """run_selfplay(compact_tail=True) with graphs and the fused tower: identical records, fewer rows evaluated."""
    from alphazero_openspiel_amd import engine as E
    from alphazero_openspiel_amd.fusednet import FusedNet
    from alphazero_openspiel_amd.network import Net
    torch.manual_seed(0)
    net = Net([3, 6, 7], 7, n_blocks=2).cuda().eval()
    G = n_games = 2048
    out = {}
    for compact in (False, True):
        eng = E.SelfPlayEngine("connect_four", G, n_playouts=12, max_games=n_games, device=0, seed=3)
        fn = FusedNet(net, "cuda:0", max_boards=G, precision=precision)
        prog = E.run_selfplay(eng, fn, n_games, use_graph=True, compact_tail=compact)
        out[compact] = (eng.export(), prog)
        eng.close()
        fn.close()
        assert prog["error_flags"] == 0 and prog["games_done"] == n_games
    _same(out[False][0], out[True][0])
    assert out[False][1]["tail_compactions"] == 0 and out[True][1]["tail_compactions"] >= 1
