#!/usr/bin/env python3
"""Runs ON THE GPU BOX: an engine-produced generation + what the device replay store makes of it, saved as a data
fixture (tests/golden/engine_export_connect_four.npz) for the container-only test that feeds the SAME records to the
real reference Trainer (tests/test_trainer_consumes_engine_output.py; the reference never travels to the GPU box).

    python tools/make_engine_export_fixture.py gpurun_out/engine_export_connect_four.npz

Contents: the packed host export of 40 self-play games (connect_four, 24 playouts/move, the reference's shipped
checkpoint evaluated by the fused HIP tower, Philox seed 123), the device store's remove_duplicates result over two
generations (24 games, then +16 with the FIFO trimmed to 32 games), one training batch gathered on the device for the
`np.random.seed(5); np.random.randint(n_unique, size=16)` draw, and the losses of one net_step on that batch (CPU fp32).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out_path):
    from alphazero_openspiel_amd import distributed as azdist, engine as E, fusednet, replay
    from alphazero_openspiel_amd.network import load_npz_checkpoint
    ckpt = os.path.join(ROOT, "tests", "golden", "checkpoint_connect_four.npz")
    net = load_npz_checkpoint(ckpt, [3, 6, 7], 7)
    fn = fusednet.FusedNet(net, "cuda:0", max_boards=32, precision="f16")
    rep = replay.DeviceReplay("connect_four", max_games=40, device=0)
    out = {}
    payloads = []
    for gen, (n_games, cap) in enumerate(((24, 40), (16, 32))):
        eng = E.SelfPlayEngine("connect_four", 32, n_playouts=24, max_games=n_games, seed=123 + gen, device=0)
        E.run_selfplay(eng, fn, n_games, use_graph=True)
        rep.set_capacity(cap)
        rep.append_engine(eng)
        payloads.append(azdist.pack_export(eng.export()))
        eng.close()
        n = rep.dedupe()
        u = rep.read_unique()
        out["gen%d_unique_pi" % gen] = u["pi"]
        out["gen%d_unique_z" % gen] = u["z"]
        out["gen%d_unique_index" % gen] = u["buffer_index"]
        out["gen%d_n_unique" % gen] = np.int64(n)
        out["gen%d_capacity" % gen] = np.int64(cap)
        out["gen%d_payload" % gen] = payloads[-1]
    np.random.seed(5)
    ids = np.random.randint(n, size=16)
    x, pi, z = rep.sample(16, indices=ids)
    out["batch_ids"], out["batch_x"], out["batch_pi"], out["batch_z"] = ids, x.cpu().numpy(), pi.cpu().numpy(), z.cpu().numpy()
    # one update with the Trainer's loss / optimiser on that batch (CPU, fp32, one thread): the reference's net_step on the
    # reference-format lists must report the same losses
    torch.set_num_threads(1)
    net_t = load_npz_checkpoint(ckpt, [3, 6, 7], 7).train()
    opt = replay.make_optimizer(net_t)
    lp, lv = replay.net_step(net_t, opt, x.cpu(), pi.cpu(), z.cpu())
    out["loss_p"], out["loss_v"] = np.float64(float(lp)), np.float64(float(lv))
    out["fc1_bias_after"] = net_t.fc1.bias.detach().numpy().astype(np.float64)
    rep.close()
    fn.close()
    np.savez_compressed(out_path, **out)
    print("wrote", out_path, os.path.getsize(out_path), "bytes;", int(n), "unique examples")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "engine_export_connect_four.npz"))
