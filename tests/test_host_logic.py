"""CPU-only tests of the host side: game objects, record -> example conversion, the façade's numpy
arithmetic, the C-ABI surface (symbols + struct layout) and the multi-rank exchange (gloo, world_size 2).
No compute entry point of the HIP library is called here."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT, load_golden
from alphazero_openspiel_amd import _lib, games
from alphazero_openspiel_amd import distributed as azdist
from alphazero_openspiel_amd.alphazerobot import remove_illegal_actions
from alphazero_openspiel_amd.engine import examples_from_export, pi_from_visits
from alphazero_openspiel_amd.network import state_to_board
from oracle import binding as orc


# ---------------------------------------------------------------------------------------------- games
@pytest.mark.parametrize("tag", ["connect_four", "breakthrough6", "breakthrough8", "breakthrough5x4"])
def test_host_games_follow_the_rule_fixtures(tag):
    blob = load_golden("rules_%s.json" % tag)
    game = games.load_game(blob["game"])
    shape = game.information_state_normalized_vector_shape()
    for g in blob["games"]:
        s = game.new_initial_state()
        for ply in g["plies"]:
            assert s.legal_actions() == ply["legal"] == s.legal_actions(s.current_player())
            assert s.current_player() == ply["player"]
            assert "".join(str(int(x)) for x in s.information_state_as_normalized_vector()) == ply["obs"]
            b = state_to_board(s, shape)
            assert b.shape == (4, game.rows, game.cols) and (b[3] == ply["player"]).all()
            bb = games.boards_from_bitboards(game, np.array([s.bb], dtype=np.uint64), [len(s.history())])[0]
            assert (bb == b).all()
            c = s.clone()
            s.apply_action(ply["action"])
            assert c.history() + [ply["action"]] == s.history()
        assert s.is_terminal() and s.returns() == g["returns"] and s.current_player() == games.TERMINAL_PLAYER
        assert s.information_state() == ", ".join(str(p["action"]) for p in g["plies"])
        with pytest.raises(ValueError):
            s.apply_action(0)


def test_game_name_parsing_and_sizes():
    g = games.load_game("breakthrough(rows=6,columns=6)")
    assert (g.rows, g.cols, g.num_distinct_actions()) == (6, 6, 432)
    assert games.load_game("breakthrough").num_distinct_actions() == 768
    assert games.load_game("connect_four").information_state_normalized_vector_shape() == [3, 6, 7]
    with pytest.raises(ValueError):
        games.load_game("chess")
    s = games.load_game("connect_four").new_initial_state()
    with pytest.raises(ValueError):
        s.apply_action(9)


# ------------------------------------------------------------------------------ record -> example logic
def test_pi_from_visits_is_the_reference_arithmetic():
    rng = np.random.RandomState(0)
    for A, nc in [(7, 7), (7, 3), (432, 17), (768, 40)]:
        for _ in range(20):
            acts = np.sort(rng.choice(A, nc, replace=False))
            visits = rng.randint(0, 400, nc).astype(np.uint32)
            visits[rng.randint(nc)] += 1
            # reference: mcts.py:161-162 then alphazerobot.py:7-18
            dense = [0] * A
            for a, v in zip(acts, visits):
                dense[a] = int(v)
            nv = np.array([float(v) / sum(dense) for v in dense])
            want = orc.remove_illegal_actions(nv, acts.tolist())
            assert pi_from_visits(acts, visits, A) == want.tolist()


def test_bulk_pi_equals_the_per_ply_arithmetic_bit_for_bit():
    """examples_from_export converts all plies of a generation in one numpy pass (pis_from_visits): every row must be the
    per-ply pi_from_visits result exactly (same IEEE operations, numpy's pairwise sum per row), ragged child counts included."""
    from alphazero_openspiel_amd.engine import pis_from_visits
    rng = np.random.RandomState(3)
    for A, mc in [(7, 7), (432, 36), (768, 48), (240, 24)]:
        n = 400
        nch = rng.randint(1, mc + 1, size=n)
        acts = np.stack([np.sort(rng.choice(A, mc, replace=False)) for _ in range(n)]).astype(np.uint16)
        vis = rng.randint(0, 2000, size=(n, mc)).astype(np.uint32)
        vis[np.arange(n), 0] += 1
        vis[::37] = 0  # no visit mass at all: the uniform-over-children fallback (alphazerobot.py:15-17)
        bulk = pis_from_visits(acts, vis, nch, A)
        for i in range(n):
            with np.errstate(all="ignore"):
                assert bulk[i].tolist() == pi_from_visits(acts[i, :nch[i]].astype(np.int64), vis[i, :nch[i]], A)


def test_remove_illegal_actions_matches_reference_fixture():
    for case in load_golden("remove_illegal.json"):
        out = remove_illegal_actions(np.array(case["probs"], dtype=np.float64), list(case["legal"]))
        assert out.tolist() == case["out"]


def _fake_export(game, n_games=3, seed=0):
    rng = np.random.RandomState(seed)
    mp, mc = game.max_game_length(), game.max_children()
    ex = {"game_len": np.zeros(n_games, np.int32), "game_ret0": np.zeros(n_games, np.float32),
          "states": np.zeros((n_games, mp, 2), np.uint64), "move": np.zeros((n_games, mp), np.uint16),
          "n_children": np.zeros((n_games, mp), np.uint8), "child_action": np.zeros((n_games, mp, mc), np.uint16),
          "child_visits": np.zeros((n_games, mp, mc), np.uint32), "value": np.zeros((n_games, mp)), "start_ply": 0}
    for g in range(n_games):
        s = game.new_initial_state()
        i = 0
        while not s.is_terminal():
            la = s.legal_actions()
            ex["states"][g, i] = s.bb
            ex["n_children"][g, i] = len(la)
            ex["child_action"][g, i, :len(la)] = la
            ex["child_visits"][g, i, :len(la)] = rng.randint(1, 50, len(la))
            a = la[rng.randint(len(la))]
            ex["move"][g, i] = a
            s.apply_action(a)
            i += 1
        ex["game_len"][g], ex["game_ret0"][g] = i, s.returns()[0]
        z = s.returns()[0]
        for j in range(i):
            ex["value"][g, j] = z
            z = -z
    return ex


@pytest.mark.parametrize("name", ["connect_four", "breakthrough(rows=6,columns=6)"])
def test_examples_have_the_reference_record_format(name):
    game = games.load_game(name)
    ex = _fake_export(game)
    out = examples_from_export(game, ex)
    assert len(out) == 3
    for g, plies in enumerate(out):
        s = game.new_initial_state()
        assert len(plies) == ex["game_len"][g]
        for i, rec in enumerate(plies):
            # what train.py needs (train.py:109-126,172-198): a mutable list of 4; key hashable; board stackable;
            # pi a python list (truthiness + zip); value a float supporting += and /
            assert isinstance(rec, list) and len(rec) == 4
            assert rec[0] == s.information_state()
            assert (rec[1] == state_to_board(s, game.information_state_normalized_vector_shape())).all()
            assert isinstance(rec[2], list) and len(rec[2]) == game.num_distinct_actions() and rec[2]
            assert abs(sum(rec[2]) - 1) < 1e-12 and all(rec[2][a] > 0 for a in s.legal_actions())
            assert isinstance(rec[3], float)
            s.apply_action(int(ex["move"][g, i]))
        assert plies[0][3] == float(ex["game_ret0"][g])
    np.array([r[1] for r in out[0]])  # stackable
    # and the multi-rank payload round-trips
    back = azdist.unpack_export(azdist.pack_export(ex))
    assert examples_from_export(game, back)[1][3][2] == out[1][3][2]


# ------------------------------------------------------------------------------------------ C ABI
def _declared_symbols():
    names = set()
    for hdr in ("az_engine.h", "az_net.h", "az_replay.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(az_[a-z_0-9]+)\s*\(", text))
    return names


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), "libaz_engine.so lacks %s declared in include/" % name
    assert declared == {n for n, _, _ in _lib.PROTOTYPES}, "ctypes prototypes out of sync with the headers"


def test_ctypes_structs_match_the_c_layout(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "az_engine.h"\n#include "az_net.h"\n#include "az_replay.h"\n'
                   'int main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(az_config), sizeof(az_sizes), '
                   'sizeof(az_progress), sizeof(az_example_view), sizeof(az_slot_info), sizeof(az_net_desc), '
                   'offsetof(az_config, seed), offsetof(az_progress, error_flags), sizeof(az_replay_config), '
                   'sizeof(az_replay_stats));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [ctypes.sizeof(_lib.AzConfig), ctypes.sizeof(_lib.AzSizes), ctypes.sizeof(_lib.AzProgress),
            ctypes.sizeof(_lib.AzExampleView), ctypes.sizeof(_lib.AzSlotInfo), ctypes.sizeof(_lib.AzNetDesc),
            _lib.AzConfig.seed.offset, _lib.AzProgress.error_flags.offset, ctypes.sizeof(_lib.AzReplayConfig),
            ctypes.sizeof(_lib.AzReplayStats)]
    assert got == want


def test_create_rejects_bad_config_without_a_gpu_call():
    lib = _lib.load()
    cfg = _lib.AzConfig()
    h = ctypes.c_void_p()
    cfg.struct_size = 1  # ABI guard trips before any HIP call
    assert lib.az_engine_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    assert b"struct_size" in lib.az_last_error(None)
    d = _lib.AzNetDesc()
    assert lib.az_net_create(ctypes.byref(d), ctypes.byref(h)) == -1


def test_engine_refuses_cpu_devices():
    from alphazero_openspiel_amd.engine import EngineError, SelfPlayEngine
    from alphazero_openspiel_amd.examplegenerator import ExampleGenerator
    from alphazero_openspiel_amd.network import Net
    with pytest.raises(EngineError):
        SelfPlayEngine("connect_four", 4, device="cpu")
    with pytest.raises(EngineError):
        ExampleGenerator(Net([3, 6, 7], 7), "connect_four", torch.device("cpu"))
    with pytest.raises(EngineError):  # nothing in the package computes on the host, whatever the options
        ExampleGenerator(Net([3, 6, 7], 7), "connect_four", torch.device("cpu"), is_test=True, generate_statistics=True)


def test_integration_md_binding_stub_matches_the_abi():
    """The ctypes stub INTEGRATION.md shows a maintainer is the struct the library checks (struct_size guard)."""
    import ctypes as C
    import re
    from alphazero_openspiel_amd import _lib
    src = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"class AzConfig\(C.Structure\):.*?\n(?=\nlib\.)", src, re.S)
    ns = {"C": C}
    exec(m.group(0), ns)
    assert list(ns["AzConfig"]._fields_) == list(_lib.AzConfig._fields_)


# ------------------------------------------------------------------------------------- multi-rank (gloo)
_WORKER = r"""
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
from alphazero_openspiel_amd import games, distributed as azdist
from alphazero_openspiel_amd.engine import examples_from_export
from alphazero_openspiel_amd.network import Net
from test_host_logic import _fake_export
dist.init_process_group("gloo")
r, w = dist.get_rank(), dist.get_world_size()
game = games.load_game("connect_four")
ex = _fake_export(game, n_games=2 + r, seed=100 + r)          # ranks hold different game counts
got = azdist.all_gather_exports(azdist.pack_export(ex))
assert len(got) == w
for k in range(w):
    want = _fake_export(game, n_games=2 + k, seed=100 + k)
    for key in ("game_len", "move", "child_visits", "states", "value"):
        n = int(want["game_len"].max())
        assert (got[k][key].reshape(len(want["game_len"]), -1)[:, :1] == want[key].reshape(len(want["game_len"]), -1)[:, :1]).all()
    assert [len(g) for g in examples_from_export(game, got[k])] == want["game_len"].tolist()
# the packed DEVICE-export layout (engine.export_device / az_engine_export_device) through the same collective:
# equal-sized buffers, one all_gather_into_tensor, sections unpack to what each rank packed
from alphazero_openspiel_amd.engine import device_export_layout, unpack_device_export
def _packed(k):
    e = _fake_export(game, n_games=3, seed=200 + k)
    layout, total = device_export_layout(3, e["move"].shape[1], e["child_action"].shape[2])
    buf = np.zeros(total, dtype=np.uint8)
    for name, dt, shape, off in layout:
        a = np.ascontiguousarray(e[name], dtype=dt).reshape(-1).view(np.uint8)
        buf[off:off + a.size] = a
    return e, buf
mine, buf = _packed(r)
allb = azdist.all_gather_device_exports(torch.from_numpy(buf)).numpy()
assert allb.size == w * buf.size
for k in range(w):
    want, _ = _packed(k)
    gotk = unpack_device_export(allb[k * buf.size:(k + 1) * buf.size], 3, want["move"].shape[1], want["child_action"].shape[2])
    for key in ("game_len", "game_ret0", "states", "move", "n_children", "child_action", "child_visits", "value"):
        assert (gotk[key] == want[key]).all(), key
    assert [len(g) for g in examples_from_export(game, gotk)] == want["game_len"].tolist()
torch.manual_seed(r)
net = Net([3, 6, 7], 7, n_blocks=2, n_filters=8)
azdist.broadcast_net(net, src=0)
flat = torch.cat([t.detach().reshape(-1).float() for t in list(net.parameters()) + list(net.buffers())])
ref = flat.clone(); dist.broadcast(ref, 0)
assert torch.equal(flat, ref)
dist.barrier(); dist.destroy_process_group()
open(os.path.join(os.environ["AZ_TEST_OUT"], "rank%%d.ok" %% r), "w").write("ok")
"""


def test_example_gather_and_weight_broadcast_world_size_2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PYTHONDONTWRITEBYTECODE="1", AZ_TEST_OUT=str(tmp_path))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29531", str(script)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert (tmp_path / "rank0.ok").exists() and (tmp_path / "rank1.ok").exists()  # (stdout of the ranks interleaves)


# ------------------------------------------------------------------------------------- round-2 host helpers
def test_slot_groups_cover_the_slots_in_multiples_of_eight():
    from alphazero_openspiel_amd.engine import slot_groups
    for G, k in ((4096, 2), (4096, 3), (100, 3), (16, 2), (5, 2), (2048, 4), (7, 1)):
        groups = slot_groups(G, k)
        assert groups[0][0] == 0 and sum(n for _, n in groups) == G and len(groups) <= k
        assert all(f == sum(n for _, n in groups[:i]) for i, (f, _) in enumerate(groups))   # contiguous
        assert all(n % 8 == 0 for _, n in groups[:-1])                                      # whole tower workgroups


def test_device_export_layout_round_trip():
    """The packed layout of az_engine_export_device (include/az_engine.h) as the host sees it: 16-byte aligned arrays in the
    order of az_example_view; unpack gives back what was packed."""
    from alphazero_openspiel_amd.engine import device_export_layout, unpack_device_export
    n, mp, mc = 5, 42, 7
    layout, total = device_export_layout(n, mp, mc)
    assert [name for name, _, _, _ in layout] == ["game_len", "game_ret0", "states", "move", "n_children", "child_action",
                                                  "child_visits", "value"]
    assert all(off % 16 == 0 for _, _, _, off in layout) and total % 16 == 0
    rng = np.random.RandomState(0)
    buf = np.zeros(total, dtype=np.uint8)
    want = {}
    for name, dt, shape, off in layout:
        a = (rng.randint(0, 200, size=shape)).astype(dt)
        want[name] = a
        raw = a.reshape(-1).view(np.uint8)
        buf[off:off + raw.size] = raw
    got = unpack_device_export(buf, n, mp, mc, start_ply=3)
    assert got["start_ply"] == 3 and all((got[k] == want[k]).all() for k in want)
    with pytest.raises(ValueError):
        unpack_device_export(buf[:-16], n, mp, mc)


def test_arena_pair_scores_follow_the_reference_convention():
    """game ids (2k, 2k+1) = one test_*_vs_* call: score1 = the first player's return with the agent first, score2 = MINUS the
    first player's return with the agent second (game_utils.py:76-82); generate_tests averages sum(score1 + score2) / (2 n)."""
    from alphazero_openspiel_amd.arena import pair_scores
    s1, s2 = pair_scores([1.0, -1.0, -1.0, -1.0, 0.0, 1.0])
    assert s1.tolist() == [1.0, -1.0, 0.0] and s2.tolist() == [1.0, 1.0, -1.0]
    assert float((s1.sum() + s2.sum()) / (2 * 3)) == pytest.approx(1.0 / 6)


def test_bench_host_core_detection_respects_the_cgroup_quota(tmp_path, monkeypatch):
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    n = bench.host_cores()
    assert 1 <= n <= (os.cpu_count() or 1)
    assert isinstance(bench.cpu_model(), str) and bench.cpu_model()
    assert bench.net_flops_per_eval(6, 7, 7, 10) == 36111600      # 36.1 MFLOP for the 10-block connect_four net
    assert abs(bench.tree_bytes_per_sim(6.5, 7, 7, 7, 6, 7) - 1722) < 1   # ... and its ~1.7 kB per sim


def test_bench_reads_hbm_traffic_only_from_a_summary_of_the_same_workload():
    """bench.py cannot collect PMC counters itself: `roofline.traffic` comes from the committed rocprofv3 summary, and only when
    that summary was taken on the workload being run (its `# workload_key:` line) - otherwise null, never another run's bytes."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    key = bench.workload_key("connect_four", 4096, 400, 10, 50, "fused", "random", "f32x", 1, 16)
    assert key.endswith("build:" + bench.build_id()) and len(bench.build_id()) == 12   # the kernels' sources are part of the key
    net, tree = bench.pmc_traffic(key)
    summary = open(os.path.join(ROOT, bench.PROFILE_SUMMARY)).read()
    if ("# workload_key: " + key) in summary:   # the committed summary was taken on THIS build of the kernels: its bytes are quoted
        assert net is not None and tree is not None
        # the tower writes its output in hi and lo halves (2 x 4096 x 42 x 64 x 2 B = 44 MB) and the head reads it back
        assert 60e6 < net < 200e6 and 5e6 < tree < 30e6
    else:                                       # a kernel source changed since: null, never the other build's bytes
        assert (net, tree) == (None, None)
    # a summary of the same workload on OTHER kernels is not quoted either
    stale = "|".join(key.split("|")[:-1] + ["build:000000000000"])
    assert bench.pmc_traffic(stale) == (None, None)
    for other in (bench.workload_key("connect_four", 4096, 400, 10, 50, "fused", "random", "f16", 1, 16),
                  bench.workload_key("connect_four", 2048, 400, 10, 50, "fused", "random", "f32x", 1, 16),
                  bench.workload_key("connect_four", 4096, 400, 10, 50, "fused", "random", "f32x", 2, 16)):
        assert bench.pmc_traffic(other) == (None, None)
