"""Search-level tolerance of the fused fp16-operand PV-net against the reference's fp32 `Net.forward`.

Same engine, same slots, same randomness (common random numbers: the device Philox streams are keyed by (seed, game id,
ply), so with one seed both runs draw the same Dirichlet vector for every move and the same uniform behind every
`np.random.choice`), two evaluators:

  * `FusedNet`           csrc/az_net.hip, fp16 MFMA operands, fp32 accumulate (the bench headline path), and, when built,
                         its split-fp16 ("f32x") variant;
  * `DeviceEvaluator`    the torch module in fp32 - reference arithmetic (network.py:48-64).  The engine driven by fp32
                         priors is bit-exact against the oracle / the reference fixtures (tests/test_engine_parity.py), so
                         this side IS the reference.

Because the draws are common, both runs of a game walk the same positions until the first move where the sampled
action differs.  On that common prefix every ply is a like-for-like comparison of one 400-playout search (with tree
reuse): L1 distance of the root visit vectors over S, argmax agreement.  After the first differing action the games
are different games, so from there on only distributions are compared: game length, outcome.

Tolerances asserted below are the ones DESIGN_HISTORY.md section 2 quotes ("within stochastic-sampling tolerance",
BASELINE.json north_star); the measured values are written to gpurun_out/precision_search_*.json.
"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu

S = 400


def _play(game_name, net, n_games, evaluator_kind, seed):
    """One generation of n_games games (all resident, no refill).  Philox streams are keyed by (seed, game id, ply), so two
    runs with the same seed see the same Dirichlet vector at every (game, ply) and the same uniform behind every choice."""
    from alphazero_openspiel_amd import engine as E, fusednet
    eng = E.SelfPlayEngine(game_name, n_games, n_playouts=S, max_games=n_games, rng="philox", seed=seed, device=0)
    if evaluator_kind == "f32":
        ev = E.DeviceEvaluator(net, "cuda:0", dtype=torch.float32)
        graph = False
    else:
        ev = fusednet.FusedNet(net, "cuda:0", max_boards=n_games, precision=evaluator_kind)
        graph = True
    prog = E.run_selfplay(eng, ev, n_games, use_graph=graph)
    assert prog["games_done"] == n_games and prog["error_flags"] == 0
    ex = eng.export()
    eng.close()
    if hasattr(ev, "close"):
        ev.close()
    return ex


def _compare(ex_a, ex_b, n_games):
    """-> statistics of run a (candidate) against run b (fp32 reference)."""
    l1, agree, same_vec, plies_cmp = [], 0, 0, 0
    first_diff, identical_games = [], 0
    for g in range(n_games):
        na, nb = int(ex_a["game_len"][g]), int(ex_b["game_len"][g])
        i = 0
        while i < min(na, nb):
            nc = int(ex_b["n_children"][g, i])
            va = ex_a["child_visits"][g, i, :nc].astype(np.int64)
            vb = ex_b["child_visits"][g, i, :nc].astype(np.int64)
            assert int(ex_a["n_children"][g, i]) == nc and (ex_a["states"][g, i] == ex_b["states"][g, i]).all()
            l1.append(np.abs(va - vb).sum() / float(vb.sum()))
            agree += int(np.argmax(va) == np.argmax(vb))
            same_vec += int((va == vb).all())
            plies_cmp += 1
            if ex_a["move"][g, i] != ex_b["move"][g, i]:
                break
            i += 1
        if i == min(na, nb) and na == nb:
            identical_games += 1
            first_diff.append(na)
        else:
            first_diff.append(i)
    la, lb = ex_a["game_len"][:n_games].astype(np.float64), ex_b["game_len"][:n_games].astype(np.float64)
    ra, rb = ex_a["game_ret0"][:n_games].astype(np.float64), ex_b["game_ret0"][:n_games].astype(np.float64)
    se = lambda x, y: float(np.sqrt(x.var() / len(x) + y.var() / len(y)) + 1e-12)
    l1 = np.array(l1)
    return {
        "games": n_games, "plies_compared": plies_cmp,
        "visit_l1_over_S_mean": float(l1.mean()), "visit_l1_over_S_p99": float(np.quantile(l1, 0.99)),
        "visit_l1_over_S_max": float(l1.max()),
        "identical_visit_vectors": same_vec / plies_cmp, "argmax_agreement": agree / plies_cmp,
        "games_identical_move_for_move": identical_games / n_games,
        "mean_plies_before_first_different_move": float(np.mean(first_diff)),
        "len_mean": [float(la.mean()), float(lb.mean())], "len_diff_in_se": float(abs(la.mean() - lb.mean()) / se(la, lb)),
        "ret0_mean": [float(ra.mean()), float(rb.mean())], "ret0_diff_in_se": float(abs(ra.mean() - rb.mean()) / se(ra, rb)),
        "p0_win_rate": [float((ra > 0).mean()), float((rb > 0).mean())], "draw_rate": [float((ra == 0).mean()), float((rb == 0).mean())],
    }


def _nets():
    from alphazero_openspiel_amd.network import Net, load_npz_checkpoint
    torch.manual_seed(1)
    yield "random10", "connect_four", Net([3, 6, 7], 7, n_blocks=10, n_filters=50).eval()
    yield "checkpoint5", "connect_four", load_npz_checkpoint(os.path.join(GOLDEN, "checkpoint_connect_four.npz"), [3, 6, 7], 7)
    yield "bt6_checkpoint5", "breakthrough(rows=6,columns=6)", load_npz_checkpoint(os.path.join(GOLDEN, "checkpoint_breakthrough6.npz"),
                                                                                   [3, 6, 6], 432)


@pytest.mark.parametrize("which", ["random10", "checkpoint5", "bt6_checkpoint5"])
def test_fused_search_stays_within_tolerance_of_fp32_reference_search(which):
    from alphazero_openspiel_amd import fusednet
    n_games = 512 if which.startswith("bt6") else 1024
    name, game_name, net = [t for t in _nets() if t[0] == which][0]
    ex32 = _play(game_name, net, n_games, "f32", seed=7)
    out = {}
    for kind in fusednet.PRECISIONS:
        out[kind] = _compare(_play(game_name, net, n_games, kind, seed=7), ex32, n_games)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "precision_search_%s.json" % which), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))
    f16 = out["f16"]
    # fp16-operand tower: per-search visit vectors within a percent or two of the fp32 search's, same preferred move.
    # Measured (round 2, 1024 games each): random-init 10-block net  mean L1/S 0.0007, argmax agreement 0.999, 98 % of the
    # games identical move for move; shipped 5-block checkpoint (sharp priors, deeper trees) 0.0085, 0.995, 80 %.
    l1_tol, agree_tol = {"random10": (0.003, 0.997), "checkpoint5": (0.02, 0.99), "bt6_checkpoint5": (0.03, 0.98)}[which]
    assert f16["visit_l1_over_S_mean"] < l1_tol and f16["argmax_agreement"] > agree_tol
    # outcome and length distributions of 1024 games indistinguishable (means within 3.5 standard errors)
    assert f16["len_diff_in_se"] < 3.5 and f16["ret0_diff_in_se"] < 3.5
    if "f32x" in out:  # split-fp16 ("fp32-grade") tower: tighter than the fp16 one on the like-for-like plies
        f32x = out["f32x"]
        assert f32x["visit_l1_over_S_mean"] <= f16["visit_l1_over_S_mean"] + 1e-9
        assert f32x["visit_l1_over_S_mean"] < 0.003 and f32x["argmax_agreement"] > 0.995
        assert f32x["len_diff_in_se"] < 3.5 and f32x["ret0_diff_in_se"] < 3.5
