"""ExampleGenerator across torch.distributed ranks on real engines: 2 processes share the one GPU of the test box,
gloo carries the generation-end exchange (on a multi-GPU node the same code runs one rank per GPU over RCCL).
Each rank plays int(n_games / world) games (examplegenerator.py:149 drops the remainder the same way) with its own
RNG streams and every rank returns the same gathered list."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

_WORKER = r"""
import hashlib, json, os, sys
import torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
from alphazero_openspiel_amd.examplegenerator import ExampleGenerator
from alphazero_openspiel_amd.network import load_npz_checkpoint
dist.init_process_group("gloo")
r = dist.get_rank()
net = load_npz_checkpoint(os.path.join(%(root)r, "tests", "golden", "checkpoint_connect_four.npz"), [3, 6, 7], 7)
if r != 0:   # the self-play ranks start from DIFFERENT (stale) weights: generate_examples must broadcast rank 0's
    with torch.no_grad():
        for p in net.parameters():
            p.add_(0.05 * torch.randn_like(p))
gen = ExampleGenerator(net, "connect_four", torch.device("cuda:0"), n_playouts=12, n_slots=8, seed=77)
games = gen.generate_examples(13)              # 13 // 2 = 6 per rank, remainder dropped
w_digest = hashlib.sha256(torch.cat([t.detach().reshape(-1).float().cpu() for t in list(gen.net.parameters()) + list(gen.net.buffers())]).numpy().tobytes()).hexdigest()
# the device path: same generator, next generation, straight into a device replay store on this rank
from alphazero_openspiel_amd.replay import DeviceReplay
rep = DeviceReplay("connect_four", max_games=64, device=0)
n_app = gen.generate_into(rep, 8)
st = rep.stats(); n_uni = rep.dedupe(); u = rep.read_unique()
r_digest = hashlib.sha256(u["pi"].tobytes() + u["z"].tobytes() + u["bitboards"].tobytes()).hexdigest()
keys = [[rec[0] for rec in g] for g in games]
digest = hashlib.sha256(json.dumps([[(rec[0], rec[2], rec[3]) for rec in g] for g in games]).encode()).hexdigest()
json.dump({"n": len(games), "digest": digest, "first_moves": [g[1][0] if len(g) > 1 else "" for g in games],
           "local_done": gen.last_progress["games_done"], "w_digest": w_digest, "n_app": n_app, "rep_games": st["n_games"],
           "n_unique": n_uni, "r_digest": r_digest},
          open(os.path.join(os.environ["AZ_TEST_OUT"], "rank%%d.json" %% r), "w"))
dist.barrier(); dist.destroy_process_group()
"""


def test_example_generator_shards_games_over_ranks(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PYTHONDONTWRITEBYTECODE="1", AZ_TEST_OUT=str(tmp_path))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29547", str(script)],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    import json
    a = json.load(open(tmp_path / "rank0.json"))
    b = json.load(open(tmp_path / "rank1.json"))
    assert a["n"] == b["n"] == 12
    assert a["w_digest"] == b["w_digest"]                   # rank 1's perturbed weights were replaced by rank 0's
    assert a["n_app"] == b["n_app"] == 8 and a["rep_games"] == b["rep_games"] == 8   # generate_into: 4 games per rank, gathered
    assert a["n_unique"] == b["n_unique"] > 8 and a["r_digest"] == b["r_digest"]     # both device stores hold the same records
    assert a["digest"] == b["digest"]                       # every rank holds the same gathered generation
    assert a["first_moves"][:6] != a["first_moves"][6:]     # the two shards are different games (different RNG streams)


_NCCL_WORKER = r"""
import json, os, sys
import torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
from alphazero_openspiel_amd import distributed as azdist
from alphazero_openspiel_amd.examplegenerator import ExampleGenerator
from alphazero_openspiel_amd.network import load_npz_checkpoint
from alphazero_openspiel_amd.replay import DeviceReplay
assert dist.get_backend() == "nccl" and not azdist._single()
net = load_npz_checkpoint(os.path.join(%(root)r, "tests", "golden", "checkpoint_connect_four.npz"), [3, 6, 7], 7)
gen = ExampleGenerator(net, "connect_four", torch.device("cuda:0"), n_playouts=12, n_slots=8, seed=77)
games = gen.generate_examples(6)            # broadcast_net + all_gather_into_tensor of the device export, over RCCL
rep = DeviceReplay("connect_four", max_games=16, device=0)
n = gen.generate_into(rep, 4)               # ... and the device-to-device append of the gathered buffer
u = rep.dedupe()
t = azdist.all_reduce_sum(torch.tensor([1.5, 2.0], dtype=torch.float64), torch.device("cuda:0"))
dist.barrier(device_ids=[0])
json.dump({"games": len(games), "appended": n, "unique": int(u), "sum": t.tolist()}, open(os.environ["AZ_TEST_OUT"], "w"))
dist.destroy_process_group()
"""


def test_rccl_code_path_in_a_one_rank_group(tmp_path):
    """The nccl (= RCCL) branch of every collective on the path - weight broadcast, all-gather of the packed device export,
    all-reduce, barrier - executed for real on the one GPU of the test box: AZ_DIST_FORCE=1 makes a one-rank process group
    run them instead of short-cutting.  (On an 8-GPU node the same calls carry 8 ranks over xGMI.)"""
    import json
    script = tmp_path / "worker.py"
    script.write_text(_NCCL_WORKER % {"root": ROOT})
    out_file = tmp_path / "out.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29553", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               AZ_DIST_FORCE="1", PYTHONDONTWRITEBYTECODE="1", AZ_TEST_OUT=str(out_file), HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    got = json.load(open(out_file))
    assert got == {"games": 6, "appended": 4, "unique": got["unique"], "sum": [1.5, 2.0]} and got["unique"] > 4


def test_bench_multi_gpu_branch_runs_over_rccl_in_a_one_rank_group(tmp_path):
    """bench.py's N > 1 branch (nccl init bound to the device, broadcast, barriers, device all-gather inside the timed region,
    MAX / SUM all-reduces) with AZ_DIST_FORCE=1 and one rank."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29557", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               AZ_DIST_FORCE="1", PYTHONDONTWRITEBYTECODE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--slots", "256",
                          "--playouts", "50", "--blocks", "2", "--cpu-baseline", "off", "--ref-seconds", "0"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["allgather_ms"] is not None and line["allgather_ms"] > 0


def test_bench_gpus_flag_starts_the_ranks_itself():
    """The driver's own form, `python bench.py --gpus N` with no torchrun environment: the parent must start N ranks as a child
    process and forward rank 0's JSON line (examplegenerator.py:140-162 shards inside one call).  Two ranks share the test box's
    one GPU (gloo transport); their aggregate must be about twice one rank's on the same small workload."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PYTHONDONTWRITEBYTECODE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    flags = ["--backend", "gloo", "--slots", "256", "--playouts", "50", "--blocks", "2", "--cpu-baseline", "off", "--ref-seconds", "0",
             "--steps", "4", "--warmup", "1"]

    def run(n):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + flags, env=env, capture_output=True,
                             text=True, timeout=900)
        assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, out.stdout[-3000:]  # rank 0 alone prints
        return json.loads(lines[0])

    two = run(2)
    assert two["n_gpus"] == 2 and two["allgather_ms"] is not None and two["allgather_ms"] > 0
    assert two["games_timed"] >= 2 * 4 * 256
    one = run(1)
    assert one["n_gpus"] == 1 and one["allgather_ms"] is None
    # two processes time-slice ONE device here (measured: 0.59 x the one-rank rate), so the aggregate cannot double; the
    # count above shows both ranks were added up, this only guards against a collapse or a double count
    assert 0.25 * one["value"] < two["value"] < 2.5 * one["value"], (one["value"], two["value"])
