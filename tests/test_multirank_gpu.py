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
gen = ExampleGenerator(net, "connect_four", torch.device("cuda:0"), n_playouts=12, n_slots=8, seed=77)
games = gen.generate_examples(13)              # 13 // 2 = 6 per rank, remainder dropped
keys = [[rec[0] for rec in g] for g in games]
digest = hashlib.sha256(json.dumps([[(rec[0], rec[2], rec[3]) for rec in g] for g in games]).encode()).hexdigest()
json.dump({"n": len(games), "digest": digest, "first_moves": [g[1][0] if len(g) > 1 else "" for g in games],
           "local_done": gen.last_progress["games_done"]},
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
    assert a["n"] == b["n"] == 12 and a["local_done"] == b["local_done"] == 6
    assert a["digest"] == b["digest"]                       # every rank holds the same gathered generation
    assert a["first_moves"][:6] != a["first_moves"][6:]     # the two shards are different games (different RNG streams)
