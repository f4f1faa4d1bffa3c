"""Fused PV-net inference (csrc/az_net.hip, C ABI include/az_net.h): host-side weight packing + handle.

What the kernel computes is exactly `Net.forward` in eval mode (reference network.py:48-64,99-104):

    x0 = board planes                                   a  = lrelu(bn1_1(x0))
    block b:  u = lrelu(bn2_b(conv1_b(a)))              -> bn2 folded into conv1:  u = lrelu(conv1'_b(a) + b1'_b)
              x = skip + conv2_b(u) + b2_b              (skip = x, or conv3_1(x0)+b3_1 for block 1)
              a = lrelu(bn1_{b+1}(x))                   -> per-channel scale/shift prologue of the next block
    logits = fc1(flatten_NCHW(x));  priors = softmax(logits[:A]);  value = tanh(logits[A])

3x3 convs are implicit GEMMs on the MFMA units: D[co][n] = sum_k Wp[co][k] * Act[k][n], n = (board, position),
k = 64 "groups" x 8 channels: group g < 63 -> (tap, c8) = divmod(g, 7), channel = 8*c8 + j, tap -> (dy,dx) =
(tap//3-1, tap%3-1); group 63 is zero padding.  So every conv is exactly 16 k-steps of 32.  Block 1's 1x1 skip
conv (conv3 on the raw planes) is not a GEMM at all: 4 input planes -> applied in fp32 into the residual stream.

Weights are stored in "fragment-linear" order: for k-step ks, output tile mt, lane l, element j:
    co = 16*mt + (l & 15),  group = 4*ks + (l >> 4)
which is the v_mfma_f32_16x16x32_f16 A-operand map, so a wave reads one contiguous KiB per fragment.
"""
import copy
import ctypes as C

import numpy as np
import torch

CPAD = 56
KSTEPS = 16
NGROUPS = 64
XOUT_C = 64
LRELU = 0.01


def _bn_affine(bn):
    s = (bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps))
    t = bn.bias.detach().double() - bn.running_mean.detach().double() * s
    return s.numpy(), t.numpy()


SPLIT_SCALE = 2048.0  # f16x3: x = hi + lo / 2048, hi = fp16(x), lo = fp16((x - hi) * 2048)


def split_fp16(a):
    """float64 array -> (hi, lo) fp16 arrays of the f16x3 representation."""
    hi = a.astype(np.float16)
    lo = ((a - hi.astype(np.float64)) * SPLIT_SCALE).astype(np.float16)
    return hi, lo


def _pack_conv(w3x3):
    """w3x3: float64 [F, Cin, 3, 3] (already folded) -> (hi, lo) fp16 bits [16, 4, 64, 8] each."""
    F_, cin = w3x3.shape[0], w3x3.shape[1]
    assert F_ <= 64 and cin <= CPAD
    dense = np.zeros((64, NGROUPS, 8), dtype=np.float64)  # [co][group][j]
    for g in range(63):
        tap, c8 = divmod(g, 7)
        ky, kx = divmod(tap, 3)
        for j in range(8):
            ci = 8 * c8 + j
            if ci < cin:
                dense[:F_, g, j] = w3x3[:, ci, ky, kx]
    out = np.zeros((KSTEPS, 4, 64, 8), dtype=np.float64)
    lane = np.arange(64)
    for ks in range(KSTEPS):
        for mt in range(4):
            out[ks, mt] = dense[16 * mt + (lane & 15), 4 * ks + (lane >> 4), :]
    hi, lo = split_fp16(out)
    return hi.view(np.uint16), lo.view(np.uint16)


def pack_net(net):
    """Net (alphazero_openspiel_amd.network.Net or the reference's Net) -> dict of packed numpy arrays."""
    if net.training:
        raise ValueError("pack_net folds BatchNorm running statistics: call net.eval() first (or pass a copy)")
    F_ = net.n_filts
    blocks = [getattr(net, "resblock%d" % (i + 1)) for i in range(getattr(net, "n_blocks", 5))]
    H, W, A = net.height, net.width, net.num_distinct_actions
    cin0 = net.num_filters_input
    if F_ > CPAD or cin0 > 4:
        raise ValueError("fused net supports n_filters <= %d and <= 4 input planes" % CPAD)
    nb = len(blocks)
    conv_w = np.zeros((2 * nb, KSTEPS, 4, 64, 8), dtype=np.uint16)
    conv_w_lo = np.zeros_like(conv_w)
    epi = np.zeros((2 * nb, 3, 64), dtype=np.float32)
    epi[:, 1, :] = 1.0
    s_in, t_in = _bn_affine(blocks[0].bn1)
    in_affine = np.zeros((2, 8), dtype=np.float32)
    in_affine[0, :cin0], in_affine[1, :cin0] = s_in, t_in
    skip_w = np.zeros((64, 4), dtype=np.float32)
    for b, blk in enumerate(blocks):
        s2, t2 = _bn_affine(blk.bn2)
        w1 = blk.conv1.weight.detach().double().numpy() * s2[:, None, None, None]
        b1 = blk.conv1.bias.detach().double().numpy() * s2 + t2
        conv_w[2 * b], conv_w_lo[2 * b] = _pack_conv(w1)
        epi[2 * b, 0, :F_] = b1
        w2 = blk.conv2.weight.detach().double().numpy()
        b2 = blk.conv2.bias.detach().double().numpy().copy()
        if blk.use_1x1conv:
            if b != 0:
                raise ValueError("a 1x1 skip conv is only supported on the first block")
            skip_w[:F_, :cin0] = blk.conv3.weight.detach().numpy()[:, :, 0, 0]
            b2 = b2 + blk.conv3.bias.detach().double().numpy()
        elif b == 0:  # identity skip on block 1 (in_planes == n_filters)
            skip_w[:F_, :cin0] = np.eye(F_, cin0, dtype=np.float32)
        conv_w[2 * b + 1], conv_w_lo[2 * b + 1] = _pack_conv(w2)
        epi[2 * b + 1, 0, :F_] = b2
        if b + 1 < nb:
            s1, t1 = _bn_affine(blocks[b + 1].bn1)
            epi[2 * b + 1, 1, :F_], epi[2 * b + 1, 2, :F_] = s1, t1
    # fc1: [A+1, F*H*W] with column index c*HW + pos  ->  Wfc'[o][pos*64 + c]
    HW = H * W
    wfc = net.fc1.weight.detach().double().numpy().reshape(A + 1, F_, HW)
    n_ot = (A + 1 + 15) // 16
    dense = np.zeros((n_ot * 16, HW, XOUT_C), dtype=np.float64)
    dense[:A + 1, :, :F_] = wfc.transpose(0, 2, 1)
    dense = dense.reshape(n_ot * 16, HW * XOUT_C)
    ksteps_fc = HW * XOUT_C // 32
    lane = np.arange(64)
    fc_w64 = np.zeros((n_ot, ksteps_fc, 64, 8), dtype=np.float64)
    kidx = (32 * np.arange(ksteps_fc)[:, None, None] + 8 * (lane >> 4)[None, :, None] + np.arange(8)[None, None, :])
    for ot in range(n_ot):
        rows = (16 * ot + (lane & 15))[None, :, None]
        fc_w64[ot] = dense[rows, kidx]
    fc_w, fc_w_lo = split_fp16(fc_w64)
    fc_b = np.zeros(n_ot * 16, dtype=np.float32)
    fc_b[:A + 1] = net.fc1.bias.detach().numpy()
    return {"conv_w": conv_w, "conv_epi": epi, "in_affine": in_affine, "skip_w": skip_w,
            "fc_w": fc_w.view(np.uint16), "fc_b": fc_b, "conv_w_lo": conv_w_lo, "fc_w_lo": fc_w_lo.view(np.uint16),
            "rows": H, "cols": W, "in_planes": cin0, "n_filters": F_, "n_blocks": nb, "num_actions": A}


def emulate_forward(packed, obs, round_fp16=True, split=False):
    """Numpy restatement of what az_net.hip does with the PACKED buffers (same group/tap/cell maps, same
    fold, optional fp16 rounding of the MFMA operands) — validates the packing on a CPU-only box.
    split=True: the f16x3 path - weights = hi + lo / 2048 from the packed pairs, activations carried as (hi, lo) pairs."""
    H, W, A, nb = packed["rows"], packed["cols"], packed["num_actions"], packed["n_blocks"]
    B = obs.shape[0]
    cells = (H + 2) * (W + 1) + 1
    rnd = (lambda a: a.astype(np.float16).astype(np.float32)) if round_fp16 else (lambda a: a.astype(np.float32))
    if split:
        def rnd(a):  # what survives the (hi, lo) representation of an activation
            hi, lo = split_fp16(a.astype(np.float64))
            return (hi.astype(np.float64) + lo.astype(np.float64) / SPLIT_SCALE).astype(np.float32)
    lane = np.arange(64)

    def unpack(cw, cw_lo=None):  # [16,4,64,8] -> dense [co 64][group 64][j 8]
        w = cw.view(np.float16).astype(np.float32)
        if cw_lo is not None:
            w = (cw.view(np.float16).astype(np.float64) + cw_lo.view(np.float16).astype(np.float64) / SPLIT_SCALE).astype(np.float32)
        dense = np.zeros((64, NGROUPS, 8), dtype=np.float32)
        for ks in range(KSTEPS):
            for mt in range(4):
                dense[16 * mt + (lane & 15), 4 * ks + (lane >> 4), :] = w[ks, mt]
        return dense

    def cell_of(y, x):
        return (y + 1) * (W + 1) + (x + 1)

    act = np.zeros((B, cells, CPAD), dtype=np.float32)
    xres = np.zeros((B, H * W, 64), dtype=np.float32)
    s_in, t_in = packed["in_affine"]
    for y in range(H):
        for x in range(W):
            v = obs[:, :, y, x].astype(np.float32)
            xres[:, y * W + x, :] = v @ packed["skip_w"][:, :v.shape[1]].T  # block-1 skip path, fp32
            a = s_in[None, :v.shape[1]] * v + t_in[None, :v.shape[1]]
            act[:, cell_of(y, x), :v.shape[1]] = rnd(np.where(a > 0, a, LRELU * a))
    for conv in range(2 * nb):
        dense = unpack(packed["conv_w"][conv], packed["conv_w_lo"][conv] if split else None)
        acc = np.zeros((B, H * W, 64), dtype=np.float32)
        for y in range(H):
            for x in range(W):
                n = y * W + x
                for g in range(63):
                    tap, c8 = divmod(g, 7)
                    dy, dx = tap // 3 - 1, tap % 3 - 1
                    src = act[:, cell_of(y, x) + dy * (W + 1) + dx, 8 * c8:8 * c8 + 8]
                    acc[:, n, :] += src @ dense[:, g, :].T
        bias, scale, shift = packed["conv_epi"][conv]
        if conv % 2 == 0:
            u = acc + bias[None, None, :]
            u = rnd(np.where(u > 0, u, LRELU * u))
            for y in range(H):
                for x in range(W):
                    act[:, cell_of(y, x), :] = u[:, y * W + x, :CPAD]
        else:
            xres = xres + acc + bias[None, None, :]
            if conv != 2 * nb - 1:
                a = scale[None, None, :] * xres + shift[None, None, :]
                a = rnd(np.where(a > 0, a, LRELU * a))
                for y in range(H):
                    for x in range(W):
                        act[:, cell_of(y, x), :] = a[:, y * W + x, :CPAD]
    tower = rnd(xres)  # [B][pos][64]
    fcw = packed["fc_w"].view(np.float16).astype(np.float32)  # [ot][ks][lane][8]
    if split:
        fcw = (packed["fc_w"].view(np.float16).astype(np.float64)
               + packed["fc_w_lo"].view(np.float16).astype(np.float64) / SPLIT_SCALE).astype(np.float32)
    n_ot, ksn = fcw.shape[0], fcw.shape[1]
    dense_fc = np.zeros((n_ot * 16, ksn * 32), dtype=np.float32)
    for ot in range(n_ot):
        for ks in range(ksn):
            for q in range(4):
                dense_fc[16 * ot + np.arange(16), 32 * ks + 8 * q:32 * ks + 8 * q + 8] = fcw[ot, ks, 16 * q:16 * q + 16, :]
    logits = tower.reshape(B, -1) @ dense_fc.T + packed["fc_b"][None, :]
    lp = logits[:, :A]
    e = np.exp(lp - lp.max(1, keepdims=True))
    return e / e.sum(1, keepdims=True), np.tanh(logits[:, A]), tower


# Arithmetic of the matrix products (include/az_net.h): "f16" = fp16 operands, one MFMA per product; "f32x" = fp32-grade,
# every operand a (hi, lo) pair of fp16 numbers, three MFMAs per product (AZ_NET_PREC_F16X3)
PRECISIONS = ("f16", "f32x")


class FusedNet:
    """Device handle: az_net_create / az_net_forward.  Call signature matches engine.DeviceEvaluator:
    evaluator(obs, priors_out, values_out)."""

    def __init__(self, net, device, max_boards=4096, precision="f32x"):
        from . import _lib
        if precision not in PRECISIONS:
            raise ValueError("precision must be one of %s" % (PRECISIONS,))
        self.precision = precision
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("FusedNet needs a HIP device; there is no CPU path")
        self.device_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", self.device_index)
        self.packed = pack_net(copy.deepcopy(net).cpu().eval())  # never move or switch the caller's module
        p = self.packed
        d = _lib.AzNetDesc()
        d.struct_size = C.sizeof(_lib.AzNetDesc)
        d.rows, d.cols, d.in_planes = p["rows"], p["cols"], p["in_planes"]
        d.n_filters, d.n_blocks, d.num_actions = p["n_filters"], p["n_blocks"], p["num_actions"]
        d.device = self.device_index
        d.precision = _lib.NET_PREC[precision]
        self._keep = [np.ascontiguousarray(p[k]) for k in ("conv_w", "conv_epi", "in_affine", "fc_w", "fc_b", "skip_w",
                                                           "conv_w_lo", "fc_w_lo")]
        d.conv_w_lo = self._keep[6].ctypes.data_as(C.POINTER(C.c_uint16))
        d.fc_w_lo = self._keep[7].ctypes.data_as(C.POINTER(C.c_uint16))
        d.conv_w = self._keep[0].ctypes.data_as(C.POINTER(C.c_uint16))
        d.conv_epi = self._keep[1].ctypes.data_as(C.POINTER(C.c_float))
        d.in_affine = self._keep[2].ctypes.data_as(C.POINTER(C.c_float))
        d.fc_w = self._keep[3].ctypes.data_as(C.POINTER(C.c_uint16))
        d.fc_b = self._keep[4].ctypes.data_as(C.POINTER(C.c_float))
        d.skip_w = self._keep[5].ctypes.data_as(C.POINTER(C.c_float))
        self._h = C.c_void_p()
        rc = self.lib.az_net_create(C.byref(d), C.byref(self._h))
        if rc != 0:
            raise RuntimeError("az_net_create failed (%d): %s" % (rc, self.lib.az_net_last_error(None).decode()))
        self._check(self.lib.az_net_reserve(self._h, int(max_boards)))
        self.max_boards = int(max_boards)
        self.A = p["num_actions"]
        self.obs_shape = (p["in_planes"], p["rows"], p["cols"])

    def _check(self, rc):
        if rc < 0:
            raise RuntimeError("fused net call failed (%d): %s" % (rc, self.lib.az_net_last_error(self._h).decode()))
        return rc

    def close(self):
        if getattr(self, "_h", None):
            self.lib.az_net_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __call__(self, obs, priors_out, values_out):
        n = obs.shape[0]
        for t, shape in ((obs, (n,) + self.obs_shape), (priors_out, (n, self.A)), (values_out, (n,))):
            if t.dtype != torch.float32 or not t.is_contiguous() or t.device != self.device or tuple(t.shape) != shape:
                raise RuntimeError("fused net: expected contiguous float32 %s on %s, got %s %s on %s"
                                   % (shape, self.device, t.dtype, tuple(t.shape), t.device))
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        self._check(self.lib.az_net_forward(self._h, C.c_void_p(obs.data_ptr()), C.c_void_p(priors_out.data_ptr()),
                                            C.c_void_p(values_out.data_ptr()), n, stream))

    def forward(self, obs):
        pri = torch.empty((obs.shape[0], self.A), dtype=torch.float32, device=self.device)
        val = torch.empty((obs.shape[0],), dtype=torch.float32, device=self.device)
        self(obs, pri, val)
        return pri, val

    def issued_mfma_per_board(self, n_boards=None):
        """v_mfma_f32_16x16x32_f16 instructions issued per board (padding included) for a launch of n_boards boards."""
        out = C.c_double()
        self._check(self.lib.az_net_issued_mfma_per_board(self._h, int(n_boards or self.max_boards), C.byref(out)))
        return out.value

    def kernel_label(self, n_boards=None):
        """Names of the kernels a forward of n_boards boards launches (default: the reserved maximum)."""
        return self.lib.az_net_kernel_label(self._h, int(n_boards or self.max_boards)).decode()

    def read_tower(self, n_boards):
        out = np.zeros((n_boards, self.packed["rows"] * self.packed["cols"], XOUT_C), dtype=np.float32)
        self._check(self.lib.az_net_read_tower(self._h, out.ctypes.data_as(C.POINTER(C.c_float)), n_boards))
        return out
