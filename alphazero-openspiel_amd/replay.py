"""Device-resident replay store + training step — the consumer side of self-play.

Reference: the Trainer's FIFO buffer of games (train.py:226-236,295-298), `remove_duplicates` (train.py:156-201) and
`net_step`'s batch sampling (train.py:107-120) run in HIP kernels on the engine's records (C ABI: include/az_replay.h),
so a generation goes  engine -> DeviceReplay.append_engine -> dedupe -> sample -> net_step  without the examples ever
becoming Python lists.  The network update (forward, MSE + cross-entropy, Adam: train.py:115-130) is plain PyTorch.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .games import Game


class DeviceReplay:
    def __init__(self, game_name, max_games, device=0, max_examples=None):
        self.lib = _lib.load()
        self.game = Game(game_name) if isinstance(game_name, str) else game_name
        dev = torch.device(device) if not isinstance(device, int) else torch.device("cuda", device)
        if dev.type != "cuda":
            raise RuntimeError("DeviceReplay lives in HBM: a HIP device is required")
        self.device = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
        cfg = _lib.AzReplayConfig()
        cfg.struct_size = C.sizeof(_lib.AzReplayConfig)
        cfg.game, cfg.rows, cfg.cols = self.game.game_id, self.game.rows, self.game.cols
        cfg.device = self.device.index
        cfg.max_games = int(max_games)
        cfg.max_examples = int(max_examples if max_examples is not None else max_games * self.game.max_game_length())
        self._h = C.c_void_p()
        rc = self.lib.az_replay_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            raise RuntimeError("az_replay_create failed (%d): %s" % (rc, self.lib.az_replay_last_error(None).decode()))
        self.A = self.game.num_distinct_actions()
        self.obs_shape = (4, self.game.rows, self.game.cols)

    def _check(self, rc):
        if rc < 0:
            raise RuntimeError("replay call failed (%d): %s" % (rc, self.lib.az_replay_last_error(self._h).decode()))
        return rc

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.az_replay_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_capacity(self, n_games):
        """Trainer.update_buffer_size (train.py:295-298): the FIFO's current size in games."""
        self._check(self.lib.az_replay_set_capacity(self._h, int(n_games)))

    def append_engine(self, engine):
        """All finished games of the engine's generation, device to device (train.py:226-227)."""
        self._check(self.lib.az_replay_append_engine(self._h, engine._h, self._stream()))

    def append_export(self, ex):
        """Games in the export layout (engine.export() / distributed.unpack_export) from host memory."""
        v = _lib.AzExampleView()
        keep = {k: np.ascontiguousarray(ex[k]) for k in ("game_len", "game_ret0", "states", "move", "n_children",
                                                         "child_action", "child_visits", "value")}
        v.n_games = len(keep["game_len"])
        v.max_plies = keep["move"].shape[1]
        v.max_children = keep["child_action"].shape[2]
        v.game_len = keep["game_len"].ctypes.data_as(C.POINTER(C.c_int32))
        v.game_ret0 = keep["game_ret0"].ctypes.data_as(C.POINTER(C.c_float))
        v.states = keep["states"].ctypes.data_as(C.POINTER(C.c_uint64))
        v.move = keep["move"].ctypes.data_as(C.POINTER(C.c_uint16))
        v.n_children = keep["n_children"].ctypes.data_as(C.POINTER(C.c_uint8))
        v.child_action = keep["child_action"].ctypes.data_as(C.POINTER(C.c_uint16))
        v.child_visits = keep["child_visits"].ctypes.data_as(C.POINTER(C.c_uint32))
        v.value = keep["value"].ctypes.data_as(C.POINTER(C.c_double))
        self._check(self.lib.az_replay_append_host(self._h, C.byref(v), int(ex.get("start_ply", 0)), self._stream()))

    def append_device(self, buf, n_games, start_ply=0):
        """Games from a packed DEVICE export (engine.export_device(), or one rank's section of the all-gathered buffer):
        engine -> RCCL all-gather -> replay store with no host copy of the records."""
        if buf.dtype != torch.uint8 or not buf.is_contiguous() or buf.device != self.device:
            raise RuntimeError("append_device expects a contiguous uint8 tensor on %s" % (self.device,))
        from .engine import device_export_layout
        need = device_export_layout(n_games, self.game.max_game_length(), self.game.max_children())[1]
        if buf.numel() < need:
            raise RuntimeError("export buffer holds %d bytes, %d games need %d" % (buf.numel(), n_games, need))
        self._check(self.lib.az_replay_append_device(self._h, C.c_void_p(buf.data_ptr()), int(n_games), int(start_ply),
                                                     self._stream()))

    def dedupe(self):
        """Trainer.remove_duplicates over the flattened buffer; returns the number of unique examples."""
        self._check(self.lib.az_replay_dedupe(self._h, self._stream()))
        return self.stats()["n_unique"]

    def stats(self):
        s = _lib.AzReplayStats()
        self._check(self.lib.az_replay_stats_get(self._h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in _lib.AzReplayStats._fields_}

    def sample(self, batch, indices=None, seed=0):
        """-> (x [B,4,H,W], pi [B,A], z [B]) float32 device tensors.  indices: int64 tensor/array of positions in the
        de-duplicated list (the reference's np.random.randint draw); None = drawn on the device."""
        x = torch.empty((batch,) + self.obs_shape, dtype=torch.float32, device=self.device)
        pi = torch.empty((batch, self.A), dtype=torch.float32, device=self.device)
        z = torch.empty((batch,), dtype=torch.float32, device=self.device)
        idx_ptr = None
        if indices is not None:
            idx = torch.as_tensor(np.asarray(indices, dtype=np.int64)).to(self.device)
            assert idx.numel() == batch
            idx_ptr = C.c_void_p(idx.data_ptr())
        self._check(self.lib.az_replay_sample(self._h, idx_ptr, int(batch), int(seed) & (2 ** 64 - 1),
                                              C.c_void_p(x.data_ptr()), C.c_void_p(pi.data_ptr()),
                                              C.c_void_p(z.data_ptr()), self._stream()))
        if indices is not None:
            torch.cuda.current_stream(self.device).synchronize()  # idx must outlive the kernel
        return x, pi, z

    def read_unique(self):
        n = self.stats()["n_unique"]
        key = np.zeros(n, np.uint64)
        pi = np.zeros((n, self.A), np.float64)
        z = np.zeros(n, np.float64)
        bidx = np.zeros(n, np.int64)
        bb = np.zeros((n, 2), np.uint64)
        ply = np.zeros(n, np.int32)
        self._check(self.lib.az_replay_read_unique(
            self._h, n, key.ctypes.data_as(C.POINTER(C.c_uint64)), pi.ctypes.data_as(C.POINTER(C.c_double)),
            z.ctypes.data_as(C.POINTER(C.c_double)), bidx.ctypes.data_as(C.POINTER(C.c_int64)),
            bb.ctypes.data_as(C.POINTER(C.c_uint64)), ply.ctypes.data_as(C.POINTER(C.c_int32))))
        return {"key": key, "pi": pi, "z": z, "buffer_index": bidx, "bitboards": bb, "ply": ply}

    def read_example(self, index):
        pi = np.zeros(self.A, np.float64)
        z = C.c_double()
        self._check(self.lib.az_replay_read_example(self._h, int(index), pi.ctypes.data_as(C.POINTER(C.c_double)),
                                                    C.byref(z)))
        return pi, z.value


def net_step(net, optimizer, x, pi_target, z_target):
    """One parameter update as Trainer.net_step does it (train.py:103,115-130): loss = MSE(v, z) +
    (-sum(pi * log p) / batch); returns (loss_p, loss_v)."""
    net.zero_grad()
    p, v = net(x)
    loss_v = torch.nn.functional.mse_loss(v, z_target.unsqueeze(1))
    loss_p = -torch.sum(pi_target * torch.log(p)) / pi_target.size(0)
    (loss_v + loss_p).backward()
    optimizer.step()
    return loss_p, loss_v


def make_optimizer(net, lr=0.001):
    """The Trainer's optimiser (train.py:86): Adam, weight decay 1e-4."""
    return torch.optim.Adam(net.parameters(), lr=lr, weight_decay=0.0001)


class GraphedNetStep:
    """net_step captured once as a HIP graph (forward, loss, backward, Adam with capturable state) and replayed per batch:
    the update is the same sequence of kernels as `net_step`, minus ~100 kernel launches' worth of host latency per step.
    Usage: step = GraphedNetStep(net, batch, replay); loss_p, loss_v = step(seed)  (samples on the device, then replays)."""

    def __init__(self, net, batch, store, lr=0.001):
        self.net, self.store, self.batch = net, store, batch
        dev = store.device
        self.opt = torch.optim.Adam(net.parameters(), lr=lr, weight_decay=0.0001, capturable=True)
        self.x = torch.zeros((batch,) + store.obs_shape, dtype=torch.float32, device=dev)
        self.pi = torch.full((batch, store.A), 1.0 / store.A, dtype=torch.float32, device=dev)
        self.z = torch.zeros((batch,), dtype=torch.float32, device=dev)
        self.loss_p = torch.zeros((), device=dev)
        self.loss_v = torch.zeros((), device=dev)
        # Warm-up outside capture (MIOpen kernel selection, allocator, Adam state creation), then put EVERYTHING it
        # touched back: weights, BatchNorm running statistics and Adam moments / step counters — the captured step
        # must be the first real update.
        import copy
        saved = copy.deepcopy(net.state_dict())
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(3):
                self._step()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        net.load_state_dict(saved)
        for st in self.opt.state.values():
            for v in st.values():
                if torch.is_tensor(v):
                    v.zero_()
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        self.opt.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph):
            self._step()

    def _step(self):
        self.opt.zero_grad(set_to_none=True)
        p, v = self.net(self.x)
        lv = torch.nn.functional.mse_loss(v, self.z.unsqueeze(1))
        lp = -torch.sum(self.pi * torch.log(p)) / self.pi.size(0)
        (lv + lp).backward()
        self.opt.step()
        self.loss_p.copy_(lp.detach())
        self.loss_v.copy_(lv.detach())

    def __call__(self, seed=0, indices=None):
        x, pi, z = self.store.sample(self.batch, indices=indices, seed=seed)
        self.x.copy_(x)
        self.pi.copy_(pi)
        self.z.copy_(z)
        self.graph.replay()
        return self.loss_p, self.loss_v
