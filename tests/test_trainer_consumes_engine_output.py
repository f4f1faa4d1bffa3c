"""The REAL reference Trainer consuming engine output unchanged (BASELINE.json north_star: "the (state, pi, z)
training-example format [is] preserved so train.py's Trainer consumes the output unchanged").

The engine only exists on the GPU box and the reference only in the build container, so the two meet through a DATA
fixture: tests/golden/engine_export_connect_four.npz, written on an MI355X by tools/make_engine_export_fixture.py - the
packed records of two engine-produced generations plus what the DEVICE replay store made of them (its
remove_duplicates result, one gathered training batch, the losses of one update on it).

Here (container only: needs /root/reference) the same records are turned into the reference's list format by
`examples_from_export`, pushed through the reference's own buffer handling (train.py:226-236), its own
`Trainer.remove_duplicates` (train.py:156-201) and its own `Trainer.net_step` (train.py:95-130), unbound, and compared:
pi / z of the de-duplicated list bit for bit (float64), the sampled batch bit for bit (float32), the losses to 2e-6.
"""
import os
import types

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import ref_harness

FIXTURE = os.path.join(GOLDEN, "engine_export_connect_four.npz")
pytestmark = [pytest.mark.reference,
              pytest.mark.skipif(not ref_harness.reference_available(), reason="needs /root/reference (build container)"),
              pytest.mark.skipif(not os.path.isfile(FIXTURE), reason="fixture not generated yet")]


def _generations():
    from alphazero_openspiel_amd import distributed as azdist, games
    from alphazero_openspiel_amd.engine import examples_from_export
    fx = np.load(FIXTURE)
    game = games.load_game("connect_four")
    gens = [examples_from_export(game, azdist.unpack_export(fx["gen%d_payload" % g])) for g in range(2)]
    return fx, game, gens


def test_engine_records_have_the_format_the_trainer_requires():
    fx, game, gens = _generations()
    assert [len(g) for g in gens] == [24, 16]
    for games_ in gens:
        for plies in games_:
            s = game.new_initial_state()
            for i, rec in enumerate(plies):
                assert type(rec) is list and len(rec) == 4                      # mutable list of 4 (train.py:177-198)
                assert rec[0] == s.information_state()                          # hashable key, equal for equal histories
                assert isinstance(rec[1], np.ndarray) and rec[1].shape == (4, 6, 7) and rec[1].dtype == np.float64
                assert type(rec[2]) is list and len(rec[2]) == 7 and all(type(v) is float for v in rec[2])
                assert bool(rec[2]) and type(rec[3]) is float                   # truthiness is tested (train.py:117,179)
                if i + 1 < len(plies):
                    s.apply_action(int(plies[i + 1][0].split(", ")[-1]))


def test_reference_remove_duplicates_on_engine_output_equals_the_device_store():
    ref = ref_harness.load_reference()
    Trainer = ref.train.Trainer
    fx, game, gens = _generations()
    buffer = []
    for g, games_ in enumerate(gens):
        for examples in games_:                      # Trainer.generate_examples (train.py:226-227)
            buffer.append(examples)
        n_games_buffer = int(fx["gen%d_capacity" % g])
        while len(buffer) > n_games_buffer:          # train.py:233-236
            del buffer[0]
        flattened = [sample for game_ in buffer for sample in game_]
        out = Trainer.remove_duplicates(flattened)   # train.py:156-201, including its write-back into `buffer`
        assert len(out) == int(fx["gen%d_n_unique" % g])
        assert [it[2] for it in out] == fx["gen%d_unique_pi" % g].tolist()
        assert [it[3] for it in out] == fx["gen%d_unique_z" % g].tolist()
        # first-occurrence order: the device store reports the flat-buffer index of every unique record
        assert [id(it) for it in out] == [id(flattened[i]) for i in fx["gen%d_unique_index" % g]]
    # ---- Trainer.net_step on the de-duplicated list: the same draw, the same batch, the same losses ----
    ckpt = os.path.join(ref_harness.REFERENCE_DIR, "models", "example_model_connect_four.pth")
    net = ref.network.Net([3, 6, 7], 7)
    net.load_state_dict(torch.load(ckpt, map_location="cpu", weights_only=True))
    net.train()
    torch.set_num_threads(1)
    np.random.seed(5)
    ids = np.random.randint(len(out), size=16)
    assert ids.tolist() == fx["batch_ids"].tolist()
    assert (np.array([out[i][1] for i in ids], dtype=np.float32) == fx["batch_x"]).all()
    assert (np.array([out[i][2] for i in ids]).astype(np.float32) == fx["batch_pi"]).all()
    assert (np.array([out[i][3] for i in ids]).astype(np.float32) == fx["batch_z"]).all()
    fake = types.SimpleNamespace(current_net=net, batch_size=16, device=torch.device("cpu"),
                                 criterion_value=torch.nn.MSELoss(), it=0,
                                 optimizer=torch.optim.Adam(net.parameters(), lr=0.001, weight_decay=0.0001))
    np.random.seed(5)
    loss_p, loss_v = Trainer.net_step(fake, out)
    assert abs(float(loss_p) - float(fx["loss_p"])) < 2e-6 and abs(float(loss_v) - float(fx["loss_v"])) < 2e-6
    assert np.abs(net.fc1.bias.detach().numpy() - fx["fc1_bias_after"]).max() < 2e-6
