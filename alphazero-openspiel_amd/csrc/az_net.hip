// az_net.hip — fused PV-net inference for gfx950 (MI355X): the whole residual tower in ONE kernel.
//
// Reference computation: Net.forward / ResidualBlock.forward (network.py:48-64,99-104) in eval mode.
// Mapping (weights pre-packed by alphazero-openspiel_amd/fusednet.py, see include/az_net.h):
//   * a wavefront owns BPW whole boards; their activations never leave the CU: one fp16 LDS image
//     [board][cell][56 ch] with a zero halo (cell = (y+1)*(W+1) + (x+1); the halo column is shared between
//     rows), rewritten in place layer after layer; the fp32 residual stream lives in registers.
//   * every 3x3 conv is an implicit GEMM on v_mfma_f32_16x16x32_f16:  D[co][n] += Wp[co][k] * Act[k][n],
//     n = (board, position) over the wave's boards, k = 64 groups x 8 channels (group -> tap, channel
//     octet; group 63 = zero padding); conv 0, which sees only the input planes, is compacted on upload to
//     16 groups (9 taps x one octet) = 4 k-steps.  M = 64 output channels (4 tiles),
//     so the accumulator of lane l holds 4 CONSECUTIVE channels of one position: the epilogue
//     (bias, LeakyReLU, next BN scale/shift) packs them to fp16 and writes 8 bytes back to the image.
//   * weights are the A operand, shared by all waves of the workgroup: streamed L2 -> LDS by
//     global_load_lds (16 B/lane) in 16 KiB chunks (4 k-steps), double buffered, one barrier per chunk.
//     They are stored fragment-linear, so an A fragment is one contiguous KiB (conflict-free ds_read_b128).
// The fc1 + softmax + tanh head is a second small MFMA kernel over the tower output (fp16, [B][HW][64]).
//
// Files: az_net_common.h (types, LDS access helpers, launch parameters), az_tower_f16.h (the kernel described above),
// az_tower_x3.h (the same tower with split-fp16 operands: fp32-grade precision), az_head.h (fc1 + softmax + tanh kernels);
// this file: the C ABI (include/az_net.h) - weight re-grouping and upload, geometry, launches.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../include/az_engine.h"
#include "../../include/az_net.h"

#include "az_net_common.h"
#include "az_head_params.h"

// ================================================================================================
struct az_net {
    az_net_desc d;
    std::string err;
    _Float16 *conv_w = nullptr, *fc_w = nullptr, *xout = nullptr;
    _Float16 *fc_w_lo = nullptr, *xout_lo = nullptr; // f16x3 only
    float *epi = nullptr, *fc_b = nullptr, *skip_w = nullptr, *logits = nullptr;
    int xc = AZ_NET_XOUT_C, fc_ksteps = 0; // channel stride of xout / k-steps of fc1 (az_net_create)
    float in_affine[16];
    int max_boards = 0;
    int bpw_max = 0, lds_head = 0, n_ot = 0, r3 = 16;
    int precision = AZ_NET_PREC_F16;
    bool x3b = false; // f16x3 on a row-pair board with <= 50 filters: the x3b scheme (no output-channel tile for channels 48, 49)
    // f16x3 with <= 50 filters on a board whose positions pack into whole column tiles (6x7, 6x6, 8x8): az_tower_x3d_kernel.
    // AZ_NET_TOWER=x3b in the environment at az_net_create keeps the kernels of round 3 (az_tower_x3b_kernel / az_tower_x3_kernel)
    // for same-box A/B runs; on row-pair boards the bits are the same either way.
    int x3d = -1;                 // variant of az_launch_tower_x3d, -1: none
    int xd_nb = 0, xd_R = 0, xd_rs = 0;
    _Float16 *conv_w_d = nullptr; // the x3d weight stream (conv_w keeps the stream of the small-batch / fallback kernel)
    uint16_t *xd_pos = nullptr, *xd_sdst = nullptr;
};
static std::string g_net_err;

#define NCHK(n, call)                                                        \
    do {                                                                     \
        hipError_t _s = (call);                                              \
        if (_s != hipSuccess) {                                              \
            (n)->err = std::string(#call) + ": " + hipGetErrorString(_s);    \
            return AZ_E_HIP;                                                 \
        }                                                                    \
    } while (0)

extern "C" const char *az_net_last_error(const az_net *n) { return n ? n->err.c_str() : g_net_err.c_str(); }

extern "C" int az_net_destroy(az_net *n) {
    if (!n) return AZ_OK;
    (void)hipSetDevice(n->d.device);
    (void)hipFree(n->conv_w);
    (void)hipFree(n->fc_w);
    (void)hipFree(n->fc_w_lo);
    (void)hipFree(n->xout_lo);
    (void)hipFree(n->xout);
    (void)hipFree(n->epi);
    (void)hipFree(n->fc_b);
    (void)hipFree(n->skip_w);
    (void)hipFree(n->logits);
    (void)hipFree(n->conv_w_d);
    (void)hipFree(n->xd_pos);
    (void)hipFree(n->xd_sdst);
    delete n;
    return AZ_OK;
}

// geometry of one launch for a given boards-per-wave
struct TowerGeom {
    int bpw, nt, ck, waves, rcells, zcell, rs, tpb, cells, off_epi, off_act, lds;
};
static TowerGeom tower_geom(int bpw, int waves, int H, int W) {
    TowerGeom g;
    g.bpw = bpw;
    if (W <= 7) { // row-pair tiles at row stride 8 (conflict-free B reads)
        g.rs = 8;
        g.tpb = (H + 1) / 2;
        g.nt = bpw * g.tpb;
    } else {
        g.rs = W + 1;
        g.tpb = 0;
        g.nt = (bpw * H * W + 15) / 16;
    }
    g.cells = (H + 2) * g.rs + 1;
    int zpad = 2 * (g.rs + 1) + 1;
    g.rcells = (bpw * g.cells + zpad + 15) & ~15;
    g.zcell = bpw * g.cells + (g.rs + 1);
    int wave_act = N_OCT * g.rcells * OCT_B;
    // waves = 8 (512 threads): small tiles only (<= 256 registers/wave), 2 waves per SIMD, so one wave's epilogue / waits
    // overlap another's MFMAs, and the eight share ONE weight stream (two co-resident 4-wave workgroups each pull their
    // own: at 1 board per wave that is ~25 TB/s of L2 reads chip-wide).
    g.waves = waves;
    int act = g.waves * wave_act;
    // 32 KiB weight chunks (half the barriers) when LDS allows; not with 8 waves: the longer unrolled body spills there
    g.ck = (act + 6144 + 2 * 8 * 4096 <= 160 * 1024) ? 8 : 4; // 32 KiB chunks (half the barriers) when LDS allows

    g.off_epi = 2 * g.ck * 4096;
    g.off_act = g.off_epi + 2048 + 8 * 64 * 8; // epilogue ring (2 KiB) + trash slots (8 B per thread, up to 512 threads)
    g.lds = g.off_act + act;
    return g;
}

// geometry of the f16x3 tower: one board per wave, 4 waves, 4-k-step chunks of (hi, lo) records
struct X3Geom {
    int nt, rcells, zcell, rs, tpb, cells, off_epi, off_act, lds, lo_off;
    bool rp1;
};
static X3Geom x3_geom(int H, int W, int r3) {
    X3Geom g;
    if (W <= 7) {
        g.rs = 8;
        g.tpb = (H + 1) / 2;
        g.nt = g.tpb;
    } else {
        g.rs = W + 1;
        g.tpb = 0;
        g.nt = (H * W + 15) / 16;
    }
    g.cells = (H + 2) * g.rs + 1;
    int zpad = 2 * (g.rs + 1) + 1;
    g.rcells = (g.cells + zpad + 15) & ~15;
    g.zcell = g.cells + (g.rs + 1);
    g.rp1 = g.tpb && g.rs == 8 && g.tpb <= 3 && g.rcells <= 96;
    const int region_b = N_OCT * g.rcells * OCT_B;
    g.lo_off = g.rp1 ? X3_LOFF_RP1 : region_b;
    const int rows = r3 < 16 ? r3 + 1 : 16, rec = 3 * 1024 + 4 * rows * 16;
    const int chunk_s = (4 * 2 * rec + 1023) & ~1023;
    g.off_epi = 2 * chunk_s;
    g.off_act = g.off_epi + 2048 + 256 * 16;
    g.lds = g.off_act + 4 * 2 * g.lo_off;
    if (g.nt < 3) g.nt = 3;
    return g;
}

// fp16 bits of conv c's weight W[co][tap][ch] in the interchange (ABI) layout: group g = tap * 7 + ch / 8, element ch % 8
static inline uint16_t abi_weight(const uint16_t *src, int c, int co, int tap, int ch) {
    const int g = tap * 7 + (ch >> 3), oks = g >> 2, olane = (g & 3) * 16 + (co & 15), mt = co >> 4;
    return src[(size_t)c * AZ_NET_KSTEPS * 2048 + ((((size_t)oks * 4 + mt) * 64 + olane) * 8) + (ch & 7)];
}

// Device weight streams of the kernels on the x3b scheme (meaning of tiles T and X: az_tower_x3b.h).  A conv is cut into parts
// (= LDS chunks): the part's k-step records [hi mt 0..2][lo mt 0..2], then the extra fragments of its k-steps in k-step order -
// conv 0: T of every k-step; conv c >= 1: T, Xhi, Xlo of k-steps 6 and 7, T of the gather k-step 14.
//   az_tower_x3b_kernel / az_tower_x3c_kernel (struct X3B): conv 0 = one part; conv c >= 1 = k-steps [0,4) [4,8) [8,12) [12,15)
//   az_tower_x3d_kernel (struct X3D):                        conv 0 = [0,2) [2,4); conv c >= 1 = [0,3) [3,6) [6,8) [8,11) [11,14) [14]
struct X3Stream {
    int n_parts0, ks0_0[2], ks1_0[2];
    size_t off_0[2], c0_b;
    int n_parts, ks0[6], ks1[6];
    size_t off[6], conv_b;
};
static X3Stream x3b_stream_layout() {
    X3Stream L = {};
    L.n_parts0 = 1, L.ks0_0[0] = 0, L.ks1_0[0] = AZ_NET_K0STEPS, L.off_0[0] = 0, L.c0_b = X3B::C0_B;
    L.n_parts = X3B::PARTS, L.conv_b = X3B::CONV_B;
    for (int part = 0; part < X3B::PARTS; part++) {
        L.ks0[part] = part * X3B::CK;
        L.ks1[part] = L.ks0[part] + X3B::CK < X3B::NKS ? L.ks0[part] + X3B::CK : X3B::NKS;
        L.off[part] = X3B::part_off(part);
    }
    return L;
}
static X3Stream x3d_stream_layout() {
    X3Stream L = {};
    L.n_parts0 = X3D::PARTS0, L.c0_b = X3D::C0_B;
    for (int part = 0; part < X3D::PARTS0; part++) L.ks0_0[part] = 2 * part, L.ks1_0[part] = 2 * part + 2, L.off_0[part] = (size_t)part * X3D::C0_PART_B;
    L.n_parts = X3D::PARTS, L.conv_b = X3D::CONV_B;
    for (int part = 0; part < X3D::PARTS; part++)
        L.ks0[part] = X3D::part_ks0(part), L.ks1[part] = X3D::part_ks0(part) + X3D::part_len(part), L.off[part] = X3D::part_off(part);
    return L;
}
// The hi halves go into these streams x 2048 (X3_WSCALE; the lo halves carry that factor in the interchange format already): with
// the activations' lo half unscaled between convs, every product of a split-fp16 multiply is then 2048 x its share and ONE
// accumulator takes all three (az_net_common.h: split_pair_planes).  Exact (a power of two) unless |w| >= 32: -> false.
static bool build_x3_stream(const X3Stream &L, const uint16_t *hi_abi, const uint16_t *lo, int n_convs, std::vector<unsigned char> &dev) {
    std::vector<uint16_t> hi_scaled((size_t)n_convs * AZ_NET_KSTEPS * 2048);
    for (size_t i = 0; i < hi_scaled.size(); i++) {
        _Float16 h;
        memcpy(&h, &hi_abi[i], 2);
        const float f = (float)h * 2048.0f;
        if (!(f > -65520.0f && f < 65520.0f)) return false;
        h = (_Float16)f;
        memcpy(&hi_scaled[i], &h, 2);
    }
    const uint16_t *hi = hi_scaled.data();
    // (+ two convs of zero padding: the kernels' fetch of "chunk + 2" is unconditional)
    dev.assign(L.c0_b + (size_t)(n_convs - 1) * L.conv_b + 2 * L.conv_b, 0);
    constexpr int FR = X3B::FR, REC2 = X3B::REC2;
    // one fragment: 64 lanes x 8 fp16; f(q, l15, j) -> bits
    auto put_frag = [&](size_t off, auto f) {
        uint16_t *o = (uint16_t *)&dev[off];
        for (int lane = 0; lane < 64; lane++)
            for (int j = 0; j < 8; j++) o[lane * 8 + j] = f(lane >> 4, lane & 15, j);
    };
    // K grouping of the 15-k-step convs: group 4 ks + q < 54 = (tap, octet) = divmod(., 6); 54, 55 zero; k-step 14 = the gather
    // k-step: element j of group q < 3 is channel 48 + (j & 1) at tap 4 q + j / 2
    auto main_val = [&](const uint16_t *src, int c, int co, int ks, int q, int j) -> uint16_t {
        if (c == 0) { // conv 0: group g < 9 = tap g of octet 0 (the input planes)
            const int g = 4 * ks + q;
            return g < 9 ? abi_weight(src, 0, co, g, j) : 0;
        }
        if (ks < 14) {
            const int gp = 4 * ks + q;
            return gp < 54 ? abi_weight(src, c, co, gp / 6, 8 * (gp % 6) + j) : 0;
        }
        const int tap = 4 * q + (j >> 1);
        return (q < 3 && tap < 9) ? abi_weight(src, c, co, tap, 48 + (j & 1)) : 0;
    };
    auto put_record = [&](size_t off, int c, int ks) { // [hi mt 0..2][lo mt 0..2]
        for (int part = 0; part < 2; part++)
            for (int mt = 0; mt < 3; mt++)
                put_frag(off + (size_t)(part * 3 + mt) * FR,
                         [&](int q, int l, int j) { return main_val(part ? lo : hi, c, 16 * mt + l, ks, q, j); });
    };
    // tile T: rows 0..3 = hi 48, hi 49, lo 48, lo 49 of a shifted-B k-step (conv 0's k-steps, the gather k-step);
    //         rows 4..7 = hi 48, hi 49, lo 48, lo 49 of the centre tap over input channels 0..47 (k-steps 6, 7)
    auto put_t = [&](size_t off, int c, int ks) {
        put_frag(off, [&](int q, int l, int j) -> uint16_t {
            const bool centre = c > 0 && (ks == 6 || ks == 7);
            const int l0 = centre ? 4 : 0;
            if (l < l0 || l >= l0 + 4) return 0;
            const uint16_t *src = (l - l0) < 2 ? hi : lo;
            const int co = 48 + ((l - l0) & 1);
            if (!centre) return main_val(src, c, co, ks, q, j);
            const int gp = 4 * ks + q; // groups 24..29 = (tap 4, octet 0..5); 30, 31 belong to tap 5: zero rows here
            return gp < 30 ? abi_weight(src, c, co, 4, 8 * (gp - 24) + j) : 0;
        });
    };
    // tile X (k-steps 6, 7): row 2 t + cc = channel 48 + cc at tap tap_of_plane(t), over input channels 0..47, unshifted B
    auto put_x = [&](size_t off, int c, int ks, const uint16_t *src) {
        put_frag(off, [&](int q, int l, int j) -> uint16_t {
            const int gp = 4 * ks + q;
            return gp < 30 ? abi_weight(src, c, 48 + (l & 1), X3B::tap_of_plane(l >> 1), 8 * (gp - 24) + j) : 0;
        });
    };
    for (int part = 0; part < L.n_parts0; part++) {
        size_t off = L.off_0[part];
        for (int ks = L.ks0_0[part]; ks < L.ks1_0[part]; ks++, off += REC2) put_record(off, 0, ks);
        for (int ks = L.ks0_0[part]; ks < L.ks1_0[part]; ks++, off += FR) put_t(off, 0, ks);
    }
    for (int c = 1; c < n_convs; c++) {
        const size_t base = L.c0_b + (size_t)(c - 1) * L.conv_b;
        for (int part = 0; part < L.n_parts; part++) {
            size_t off = base + L.off[part];
            for (int ks = L.ks0[part]; ks < L.ks1[part]; ks++, off += REC2) put_record(off, c, ks);
            for (int ks = L.ks0[part]; ks < L.ks1[part]; ks++) {
                if (ks == 6 || ks == 7) {
                    put_t(off, c, ks);
                    put_x(off + FR, c, ks, hi);
                    put_x(off + 2 * FR, c, ks, lo);
                    off += 3 * FR;
                } else if (ks == X3B::NKS - 1) {
                    put_t(off, c, ks);
                    off += FR;
                }
            }
        }
    }
    return true;
}

// Column layout of az_tower_x3d_kernel (az_tower_x3d.h) for an H x W board: boards per workgroup, row stride, cells per board
// region, and the tables - which (board, position) is column 16 k + l15 of tile k, and for every column and tap plane the column
// that takes its tile-X term.  Board b's cells: b R + (y + 1) rs + x + 1.  R > H rs + W keeps every tap of every position inside
// the board or on a halo cell; (R, rs) - constants of the kernel variant - make the nb H W cells fall into the 16 residues mod 16
// equally often; tile k takes the k-th position (in board, position order) of every residue, lane l15 the one of residue l15.
struct X3DLayout {
    int variant = -1, nb = 0, R = 0, rs = 0;
    std::vector<uint16_t> pos, sdst;
};
static X3DLayout x3d_layout(int H, int W) {
    X3DLayout out;
    const int HW = H * W;
    for (int v = 0; v < AZ_X3D_VARIANTS && out.variant < 0; v++) {
        const X3DVariant V = az_x3d_variant(v);
        const int ncol = 16 * V.tiles;
        if (ncol % HW) continue;
        const int nb = ncol / HW;
        // (rs, R) are compile-time constants of the kernel variant: check that they suit THIS board - every tap of every position
        // inside the board or on a halo cell, the planes large enough, the residues flat
        const int rs = V.rs, R = V.R;
        if (rs < W + 1 || R < H * rs + W + 1 || (nb - 1) * R + (H + 1) * rs + W + 2 > V.pc) continue;
        int hist[16] = {0};
        for (int b = 0; b < nb; b++)
            for (int pp = 0; pp < HW; pp++) hist[(b * R + (pp / W + 1) * rs + pp % W + 1) & 15]++;
        bool flat = true;
        for (int r = 0; r < 16; r++) flat = flat && hist[r] == V.tiles;
        if (flat) out.variant = v, out.nb = nb, out.R = R, out.rs = rs;
    }
    if (out.variant < 0) return out;
    const X3DVariant V = az_x3d_variant(out.variant);
    const int ncol = 16 * V.tiles;
    out.pos.assign(ncol, 0);
    std::vector<int> col_of((size_t)out.nb * HW, -1), seen(16, 0);
    for (int b = 0; b < out.nb; b++)
        for (int pp = 0; pp < HW; pp++) {
            const int r = (b * out.R + (pp / W + 1) * out.rs + pp % W + 1) & 15, k = seen[r]++;
            out.pos[k * 16 + r] = (uint16_t)(b << 8 | pp);
            col_of[(size_t)b * HW + pp] = k * 16 + r;
        }
    out.sdst.assign((size_t)ncol * 8, 0xFFFF);
    for (int col = 0; col < ncol; col++) {
        const int b = out.pos[col] >> 8, pp = out.pos[col] & 255, y = pp / W, x = pp % W;
        for (int t = 0; t < 8; t++) { // plane t = tap tap_of_plane(t) with d = (dy, dx): this column's value is a term of out[position - d]
            const int tap = X3D::tap_of_plane(t), dy = tap / 3 - 1, dx = tap % 3 - 1, yd = y - dy, xd = x - dx;
            if (yd >= 0 && yd < H && xd >= 0 && xd < W) out.sdst[(size_t)col * 8 + t] = (uint16_t)col_of[(size_t)b * HW + yd * W + xd];
        }
    }
    return out;
}

#define AZ_X3C_MAX_BOARDS 512 // (set from profiles/r3_tower_vs_boards.txt)
#define AZ_F16C_MAX_BOARDS 512 // the same for the f16 tower (az_tower_f16c.h)
#define AZ_X3C_ONE_PER_WG 256 // up to here a board per workgroup fills fewer CUs than the chip has; above, two boards per workgroup (39 vs 49 us at 512 boards)
// v_mfma instructions one wave (= one board) of az_tower_x3_kernel issues (az_tower_x3.h: 3 per product, every tile)
static double x3_mfma_per_wave(int nt, int n_convs, int nks) { return 3.0 * (AZ_NET_K0STEPS + (double)(n_convs - 1) * nks) * 4 * nt; }

extern "C" int az_net_create(const az_net_desc *desc, az_net **out) {
    if (!desc || !out) {
        g_net_err = "null argument";
        return AZ_E_INVALID;
    }
    *out = nullptr;
    if (desc->struct_size != (int32_t)sizeof(az_net_desc)) {
        g_net_err = "az_net_desc.struct_size mismatch";
        return AZ_E_INVALID;
    }
    const az_net_desc &d = *desc;
    if (d.rows < 3 || d.cols < 3 || d.rows * d.cols > 64 || d.in_planes < 1 || d.in_planes > 4 || d.n_filters < 1 ||
        d.n_filters > AZ_NET_CPAD || d.n_blocks < 1 || d.num_actions < 1 || !d.conv_w || !d.conv_epi || !d.in_affine ||
        !d.skip_w || !d.fc_w || !d.fc_b) {
        g_net_err = "bad net description (need 3<=rows,cols, rows*cols<=64, in_planes<=4, n_filters<=56, packed buffers)";
        return AZ_E_INVALID;
    }
    if (d.precision != AZ_NET_PREC_F16 && d.precision != AZ_NET_PREC_F16X3) {
        g_net_err = "precision must be AZ_NET_PREC_F16 or AZ_NET_PREC_F16X3";
        return AZ_E_INVALID;
    }
    if (d.precision == AZ_NET_PREC_F16X3 && (!d.conv_w_lo || !d.fc_w_lo)) {
        g_net_err = "AZ_NET_PREC_F16X3 needs conv_w_lo and fc_w_lo";
        return AZ_E_INVALID;
    }
    az_net *n = new az_net();
    n->d = d;
    n->precision = d.precision;
    memcpy(n->in_affine, d.in_affine, sizeof n->in_affine);
    const int HW = d.rows * d.cols;
    if (n->precision == AZ_NET_PREC_F16X3) {
        X3Geom g = x3_geom(d.rows, d.cols, d.n_filters <= 50 ? 2 : 16);
        n->x3b = g.rp1 && d.n_filters <= 50; // row-pair board, channels 48, 49 the only ones past three tiles
        if (!n->x3b && (g.nt > 4 || g.lds > 160 * 1024)) {
            g_net_err = "board / filter count does not fit the f16x3 tower kernel's LDS budget";
            delete n;
            return AZ_E_INVALID;
        }
    }
    // boards per wave: at most 4 column tiles per wave (larger tiles spill registers under the hand-scheduled k-loop
    // and measured slower than more, smaller waves) within the 160 KiB LDS
    int best = 0;
    for (int bpw = 1; bpw <= 8; bpw++) {
        TowerGeom g = tower_geom(bpw, 4, d.rows, d.cols);
        if (g.nt > 4 || g.lds > 160 * 1024) break;
        best = bpw;
    }
    if (!best) {
        g_net_err = "board does not fit the tower kernel's LDS budget";
        delete n;
        return AZ_E_INVALID;
    }
    n->bpw_max = best;
    n->n_ot = (d.num_actions + 1 + 15) / 16;
    n->lds_head = HEAD_NW * OTG * 64 * 16 + 16 * n->n_ot * 16 * 4;
    hipError_t s = hipSetDevice(d.device);
    if (s != hipSuccess) {
        g_net_err = std::string("hipSetDevice: ") + hipGetErrorString(s);
        delete n;
        return AZ_E_HIP;
    }
    // Channel stride of the tower output = K index of fc1.  The descriptor's fc stream is laid out for 64 (AZ_NET_XOUT_C: 50 channels
    // padded to two 32-wide k-steps per cell).  Where fc1 is a real GEMM (az_head_gemm_kernel) the 14 padding channels are 22 % of its
    // MFMAs and bytes: those nets keep 52 channels per cell (8-byte stores stay aligned; H*W even keeps the board rows 16-byte aligned)
    // and the stream is repacked here.  The small heads (az_head_kernel, az_head_fused.h) walk k-steps of (cell, channel half): 64.
    n->xc = (n->n_ot > OTG && HW % 2 == 0 && d.n_filters <= 52) ? 52 : AZ_NET_XOUT_C; // (the reference's nets have 50 filters)
    n->fc_ksteps = (HW * n->xc + 31) / 32;
    const int ks64 = HW * AZ_NET_XOUT_C / 32;
    size_t fw = (size_t)n->n_ot * n->fc_ksteps * 64 * 8 * 2, fb = (size_t)n->n_ot * 16 * 4;
    std::vector<uint16_t> fc_re, fc_re_lo;
    auto repack_fc = [&](const uint16_t *src, std::vector<uint16_t> &dst) { // [ot][k-step][lane][8]: k = 32 ks + 8 (lane >> 4) + e, row lane & 15
        dst.assign(fw / 2, 0);
        for (int ot = 0; ot < n->n_ot; ot++)
            for (int ks = 0; ks < n->fc_ksteps; ks++)
                for (int lane = 0; lane < 64; lane++)
                    for (int e = 0; e < 8; e++) {
                        const int k = 32 * ks + 8 * (lane >> 4) + e;
                        if (k >= HW * n->xc) continue;
                        const int k64 = (k / n->xc) * AZ_NET_XOUT_C + k % n->xc;
                        dst[(((size_t)ot * n->fc_ksteps + ks) * 64 + lane) * 8 + e] =
                            src[(((size_t)ot * ks64 + k64 / 32) * 64 + ((k64 % 32) / 8) * 16 + (lane & 15)) * 8 + k64 % 8];
                    }
    };
    int rc = AZ_OK;
    auto up = [&](void **dst, const void *src, size_t bytes) {
        if (rc != AZ_OK) return;
        if (hipMalloc(dst, bytes) != hipSuccess || hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) != hipSuccess) {
            g_net_err = "hipMalloc/hipMemcpy of the packed weights failed";
            rc = AZ_E_NOMEM;
        }
    };
    {   // device layout: [conv 0 compacted to AZ_NET_K0STEPS k-steps][conv 1 ..][..], one RECORD per k-step: output-channel
        // tiles 0..2 as in the ABI layout (3 KiB), tile 3 with only its stored rows (see WRec).  Conv 0 sees the input
        // planes only (channel octet 0), i.e. ABI groups 7*tap; they become groups 0..8 of its 4 k-steps.
        // <= 50 filters: the K dimension is re-grouped into 15 k-steps (see dev_group below) - channels 48, 49 of the nine
        // taps fit ONE k-step instead of filling the seventh channel octet of every tap.
        n->r3 = d.n_filters <= 50 ? 2 : 16;
        const bool l15 = n->r3 < 16;
        const int nks = l15 ? 15 : AZ_NET_KSTEPS;
        const int rows = n->r3 < 16 ? n->r3 + 1 : 16, rec = 3 * 1024 + 4 * rows * 16;
        const int n_convs = 2 * d.n_blocks;
        const size_t conv_b = (size_t)AZ_NET_KSTEPS * 4096;
        const size_t n_rec = (size_t)AZ_NET_K0STEPS + (size_t)(n_convs - 1) * nks;
        auto build_records = [&](const unsigned char *src, std::vector<unsigned char> &dev) { // one `rec`-byte record per k-step
        dev.assign(n_rec * rec, 0);
        auto put_record = [&](unsigned char *dst, const unsigned char *ks4k) { // ks4k: [4 mt][64 lanes][16 B]
            memcpy(dst, ks4k, 3 * 1024);
            for (int q = 0; q < 4; q++)
                for (int r = 0; r < (n->r3 < 16 ? n->r3 : 16); r++)
                    memcpy(dst + 3 * 1024 + (q * rows + r) * 16, ks4k + 3 * 1024 + (q * 16 + r) * 16, 16);
        };
        // ABI: element j of group g = tap * 7 + c8 is channel 8 * c8 + j at that tap.  abi_half: one fp16 of a conv
        auto abi_half = [&](int c, int mt, int l15_, int tap, int ch) -> uint16_t {
            int g = tap * 7 + (ch >> 3), oks = g >> 2, olane = (g & 3) * 16 + l15_;
            const uint16_t *w = (const uint16_t *)(src + (size_t)c * conv_b);
            return w[((((size_t)oks * 4 + mt) * 64 + olane) * 8) + (ch & 7)];
        };
        std::vector<unsigned char> c0(AZ_NET_K0STEPS * 4096, 0); // conv 0 compacted, still in 4 KiB k-steps
        for (int ks = 0; ks < AZ_NET_K0STEPS; ks++)
            for (int mt = 0; mt < 4; mt++)
                for (int lane = 0; lane < 64; lane++) {
                    int g = 4 * ks + (lane >> 4);
                    if (g >= 9) continue;
                    int go = 7 * g, oks = go >> 2, olane = (go & 3) * 16 + (lane & 15);
                    memcpy(&c0[(((size_t)ks * 4 + mt) * 64 + lane) * 16], src + (((size_t)oks * 4 + mt) * 64 + olane) * 16, 16);
                }
        size_t off = 0;
        for (int ks = 0; ks < AZ_NET_K0STEPS; ks++, off += rec) put_record(&dev[off], &c0[(size_t)ks * 4096]);
        std::vector<uint16_t> k4(4096 / 2);
        for (int c = 1; c < n_convs; c++)
            for (int ks = 0; ks < nks; ks++, off += rec) {
                if (!l15) {
                    put_record(&dev[off], src + (size_t)c * conv_b + (size_t)ks * 4096);
                    continue;
                }
                // 15-k-step grouping: group g' = 4 ks + q.  g' < 54: (tap, octet) = divmod(g', 6), the 48 channels of six
                // full octets; g' = 54, 55: zero; k-step 14: element j of group q < 3 is channel 48 + (j & 1) at tap
                // 4 q + j / 2 (taps > 8: zero), group 3 zero.
                std::fill(k4.begin(), k4.end(), (uint16_t)0);
                for (int mt = 0; mt < 4; mt++)
                    for (int lane = 0; lane < 64; lane++) {
                        const int q = lane >> 4, l = lane & 15, gp = 4 * ks + q;
                        uint16_t *o = &k4[(((size_t)mt * 64) + lane) * 8];
                        for (int j = 0; j < 8; j++) {
                            if (ks < 14) {
                                if (gp < 54) o[j] = abi_half(c, mt, l, gp / 6, 8 * (gp % 6) + j);
                            } else if (q < 3) {
                                int tap = 4 * q + (j >> 1);
                                if (tap < 9) o[j] = abi_half(c, mt, l, tap, 48 + (j & 1));
                            }
                        }
                    }
                put_record(&dev[off], (const unsigned char *)k4.data());
            }
        };
        std::vector<unsigned char> hi, dev;
        const char *tw = getenv("AZ_NET_TOWER");
        if (n->precision == AZ_NET_PREC_F16X3 && d.n_filters <= 50 && !(tw && strcmp(tw, "x3b") == 0)) {
            const X3DLayout L = x3d_layout(d.rows, d.cols);
            if (L.variant >= 0) {
                n->x3d = L.variant, n->xd_nb = L.nb, n->xd_R = L.R, n->xd_rs = L.rs;
                std::vector<unsigned char> sd;
                if (!build_x3_stream(x3d_stream_layout(), d.conv_w, d.conv_w_lo, n_convs, sd)) {
                    g_net_err = "AZ_NET_PREC_F16X3: a conv weight of magnitude >= 32 (the device copy holds the weights x 2048 in fp16)";
                    rc = AZ_E_INVALID;
                }
                up((void **)&n->conv_w_d, sd.data(), sd.size());
                up((void **)&n->xd_pos, L.pos.data(), L.pos.size() * 2);
                up((void **)&n->xd_sdst, L.sdst.data(), L.sdst.size() * 2);
            }
        }
        if (n->x3b) {
            if (!build_x3_stream(x3b_stream_layout(), d.conv_w, d.conv_w_lo, n_convs, dev)) {
                g_net_err = "AZ_NET_PREC_F16X3: a conv weight of magnitude >= 32 (the device copy holds the weights x 2048 in fp16)";
                rc = AZ_E_INVALID;
            }
            up((void **)&n->conv_w, dev.data(), dev.size());
        } else {
        build_records((const unsigned char *)d.conv_w, hi);
        const size_t pad = 2 * 8 * 4096 + 1024; // a chunk of padding: the last (short) chunk is fetched at full length
        if (n->precision == AZ_NET_PREC_F16X3) { // per k-step: hi record, lo record
            std::vector<unsigned char> lo;
            build_records((const unsigned char *)d.conv_w_lo, lo);
            dev.assign(2 * n_rec * rec + pad, 0);
            for (size_t r = 0; r < n_rec; r++) {
                memcpy(&dev[2 * r * rec], &hi[r * rec], rec);
                memcpy(&dev[(2 * r + 1) * rec], &lo[r * rec], rec);
            }
        } else {
            dev = hi;
            dev.resize(n_rec * rec + pad, 0);
        }
        up((void **)&n->conv_w, dev.data(), dev.size());
        }
    }
    {   // [conv][3][64] (ABI) -> [conv][4][64] with the NEXT conv's bias in row 3 (what the kernel's ring slot holds; f32x: x 2048)
        int nc = 2 * d.n_blocks;
        std::vector<float> e4((size_t)nc * 256, 0.f);
        for (int c = 0; c < nc; c++) {
            memcpy(&e4[(size_t)c * 256], d.conv_epi + (size_t)c * 192, 192 * sizeof(float));
            if (c + 1 < nc) memcpy(&e4[(size_t)c * 256 + 192], d.conv_epi + (size_t)(c + 1) * 192, 64 * sizeof(float));
            // fp32-grade towers: an accumulator holds 2048 x the conv (X3_WSCALE), so its initial value is 2048 x the bias - scaled here
            // (exact), not by two multiplies per output tile in every epilogue
            if (n->precision == AZ_NET_PREC_F16X3)
                for (int i = 0; i < 64; i++) e4[(size_t)c * 256 + 192 + i] *= X3_WSCALE;
        }
        up((void **)&n->epi, e4.data(), e4.size() * sizeof(float));
    }
    if (n->xc != AZ_NET_XOUT_C) {
        repack_fc(d.fc_w, fc_re);
        up((void **)&n->fc_w, fc_re.data(), fw);
        if (n->precision == AZ_NET_PREC_F16X3) {
            repack_fc(d.fc_w_lo, fc_re_lo);
            up((void **)&n->fc_w_lo, fc_re_lo.data(), fw);
        }
    } else {
        up((void **)&n->fc_w, d.fc_w, fw);
        if (n->precision == AZ_NET_PREC_F16X3) up((void **)&n->fc_w_lo, d.fc_w_lo, fw);
    }
    up((void **)&n->fc_b, d.fc_b, fb);
    up((void **)&n->skip_w, d.skip_w, 64 * 4 * sizeof(float));
    if (rc != AZ_OK) {
        az_net_destroy(n);
        return rc;
    }
    n->d.conv_w_lo = nullptr;
    n->d.fc_w_lo = nullptr;
    n->d.conv_w = nullptr; // host pointers are not kept
    n->d.conv_epi = nullptr;
    n->d.in_affine = nullptr;
    n->d.skip_w = nullptr;
    n->d.fc_w = nullptr;
    n->d.fc_b = nullptr;
    *out = n;
    return AZ_OK;
}

extern "C" int az_net_reserve(az_net *n, int32_t max_boards) {
    if (!n || max_boards < 1) return AZ_E_INVALID;
    NCHK(n, hipSetDevice(n->d.device));
    if (n->xout) (void)hipFree(n->xout);
    n->xout = nullptr;
    // (+ 64 bytes: the last k-step of fc1 may reach past a board's row - into the next board's, times zero weights)
    size_t bytes = (size_t)max_boards * n->d.rows * n->d.cols * n->xc * 2 + 64;
    NCHK(n, hipMalloc((void **)&n->xout, bytes));
    NCHK(n, hipMemset(n->xout, 0, bytes));
    if (n->xout_lo) (void)hipFree(n->xout_lo);
    n->xout_lo = nullptr;
    if (n->precision == AZ_NET_PREC_F16X3) {
        NCHK(n, hipMalloc((void **)&n->xout_lo, bytes));
        NCHK(n, hipMemset(n->xout_lo, 0, bytes));
    }
    if (n->logits) (void)hipFree(n->logits);
    n->logits = nullptr;
    if (n->n_ot > OTG) NCHK(n, hipMalloc((void **)&n->logits, (size_t)max_boards * n->n_ot * 16 * sizeof(float) * 2)); // x 2: HG_KSPLIT partial sums (az_head_gemm_kernel)
    n->max_boards = max_boards;
    return AZ_OK;
}

// Work partition of the f16 tower for THIS batch size.  Candidates: boards per wave x {4, 8} waves per workgroup.  One
// workgroup per CU is resident, a launch runs in ceil(WGs / 256) rounds and a round costs ~ (2 * tiles + 1), x1.5 when two
// waves share each SIMD: pick the candidate that minimises rounds x round cost.
static TowerGeom choose_geom(const az_net *n, int n_boards) {
    TowerGeom g = tower_geom(1, 4, n->d.rows, n->d.cols);
    double best_cost = -1;
    for (int bpw = 1; bpw <= n->bpw_max; bpw++)
        for (int waves = 4; waves <= 8; waves += 4) {
            TowerGeom c = tower_geom(bpw, waves, n->d.rows, n->d.cols);
            if (c.lds > 160 * 1024 || (waves == 8 && c.nt > 3)) continue; // 8 waves need <= 256 registers each
            long wgs = (n_boards + waves * bpw - 1) / (waves * bpw);
            long rounds = (wgs + 255) / 256;
            double cost = rounds * (2.0 * (c.nt < 3 ? 3 : c.nt) + 1.0) * (waves == 8 ? 1.5 : 1.0);
            if (best_cost < 0 || cost < best_cost) {
                best_cost = cost;
                g = c;
            }
        }
    return g;
}

// which kernels az_net_forward launches for a batch of n_boards (the SAME conditions as there)
struct NetDispatch {
    bool x3c, x3c_head, x3d, f16c;
};
static NetDispatch net_dispatch(const az_net *n, int n_boards) {
    NetDispatch d = {false, false, false, false};
    const int ksteps = n->fc_ksteps;
    if (n->precision == AZ_NET_PREC_F16X3) {
        d.x3c = n->x3b && n_boards <= AZ_X3C_MAX_BOARDS;
        d.x3c_head = d.x3c && n->n_ot == 1 && ksteps <= 96 && n->d.cols >= 4;
        // packed tiles put xd_nb boards in a workgroup: with eight (6x6) a batch must be large enough to occupy the chip - at 1024
        // boards 128 workgroups of 4.5 tiles per SIMD lose to 256 of 3 (az_tower_x3b_kernel, one round), from 1280 on they win
        d.x3d = !d.x3c && n->x3d >= 0 && (!n->x3b || n_boards > 128 * n->xd_nb);
    } else {
        const TowerGeom gc = tower_geom(1, 1, n->d.rows, n->d.cols);
        d.f16c = n_boards <= AZ_F16C_MAX_BOARDS && n->r3 == 2 && gc.tpb && gc.rs == 8 && gc.tpb <= 3;
    }
    return d;
}

extern "C" int az_net_issued_mfma_per_board(const az_net *n, int32_t n_boards, double *out) {
    if (!n || !out || n_boards < 1) return AZ_E_INVALID;
    const int n_convs = 2 * n->d.n_blocks, nks = n->r3 < 16 ? 15 : AZ_NET_KSTEPS;
    const NetDispatch dp = net_dispatch(n, n_boards);
    const double head = (double)n->n_ot * n->fc_ksteps / 16.0; // one MFMA per (output tile, k-step) per 16 boards
    // a column tile on the x3b scheme: conv 0: 4 k-steps x (9 + 2 T); then 15 x 9 + 2 x (2 T + 3 X) + 2 T (gather k-step)
    const double per_tile = AZ_NET_K0STEPS * 11 + (double)(n_convs - 1) * (15 * 9 + 2 * 5 + 2);
    if (n->precision == AZ_NET_PREC_F16X3) {
        const X3Geom g = x3_geom(n->d.rows, n->d.cols, n->r3);
        if (dp.x3d) *out = az_x3d_variant(n->x3d).tiles * per_tile / n->xd_nb + 3.0 * head; // the workgroup's tiles over its boards
        else if (n->x3b) // three tiles per board (x3b, x3c); the head fused into x3c runs its eight chains once per board or pair of boards
            *out = 3.0 * per_tile + 3.0 * (dp.x3c_head ? head * 16.0 / (n_boards > AZ_X3C_ONE_PER_WG ? 2 : 1) : head);
        else
            *out = x3_mfma_per_wave(g.nt <= 3 ? 3 : 4, n_convs, nks) + 3.0 * head; // one board per wave
        return AZ_OK;
    }
    if (dp.f16c) { // one board per four-wave workgroup: three column tiles x 4 output-channel tiles
        *out = (double)(AZ_NET_K0STEPS + (n_convs - 1) * nks) * 4 * 3 + head;
        return AZ_OK;
    }
    const TowerGeom g = choose_geom(n, n_boards);
    const int nt = g.nt < 3 ? 3 : (g.nt > 3 ? 4 : 3);
    *out = (double)(AZ_NET_K0STEPS + (n_convs - 1) * nks) * 4 * nt / g.bpw + head;
    return AZ_OK;
}

extern "C" const char *az_net_kernel_label(const az_net *n, int32_t n_boards) {
    if (!n) return "";
    const bool big = n->n_ot > OTG;
    const NetDispatch dp = net_dispatch(n, n_boards < 1 ? n->max_boards : n_boards);
    if (n->precision == AZ_NET_PREC_F16X3) {
        if (dp.x3c_head) return "az_tower_x3c_kernel (fc1 + softmax + tanh in the same launch)";
        if (dp.x3c) return big ? "az_tower_x3c_kernel + az_head_gemm_kernel<X3> + az_head_softmax_kernel<X3>" : "az_tower_x3c_kernel + az_head_kernel<X3>";
        if (dp.x3d) return big ? "az_tower_x3d_kernel + az_head_gemm_kernel<X3> + az_head_softmax_kernel<X3>" : "az_tower_x3d_kernel + az_head_kernel<X3>";
        if (n->x3b) return big ? "az_tower_x3b_kernel + az_head_gemm_kernel<X3> + az_head_softmax_kernel<X3>" : "az_tower_x3b_kernel + az_head_kernel<X3>";
        return big ? "az_tower_x3_kernel + az_head_gemm_kernel<X3> + az_head_softmax_kernel<X3>" : "az_tower_x3_kernel + az_head_kernel<X3>";
    }
    if (dp.f16c) return big ? "az_tower_f16c_kernel + az_head_gemm_kernel + az_head_softmax_kernel" : "az_tower_f16c_kernel + az_head_kernel";
    return big ? "az_tower_kernel + az_head_gemm_kernel + az_head_softmax_kernel" : "az_tower_kernel + az_head_kernel";
}

extern "C" int az_net_forward(az_net *n, const float *obs, float *priors, float *values, int32_t n_boards, void *stream) {
    if (!n || !obs || !priors || !values || n_boards < 1) return AZ_E_INVALID;
    if (n_boards > n->max_boards) {
        n->err = "n_boards exceeds az_net_reserve()";
        return AZ_E_STATE;
    }
    NCHK(n, hipSetDevice(n->d.device)); // the launch must pair `stream` with the device the net lives on
    hipStream_t st = (hipStream_t)stream;
    HeadParams hp;
    hp.HW = n->d.rows * n->d.cols;
    hp.A = n->d.num_actions;
    hp.n_ot = n->n_ot;
    hp.K = hp.HW * n->xc;
    hp.ksteps = n->fc_ksteps;
    hp.n_boards = n_boards;
    hp.x = n->xout;
    hp.x_lo = n->xout_lo;
    hp.fc_w = n->fc_w;
    hp.fc_w_lo = n->fc_w_lo;
    hp.fc_b = n->fc_b;
    hp.priors = priors;
    hp.values = values;
    if (n->precision == AZ_NET_PREC_F16X3) { // split-fp16 tower: one board per wave, 4 waves per workgroup
        const X3Geom g = x3_geom(n->d.rows, n->d.cols, n->r3);
        TowerParams tp;
        tp.H = n->d.rows;
        tp.W = n->d.cols;
        tp.HW = tp.H * tp.W;
        tp.cells = g.cells;
        tp.rs = g.rs;
        tp.tpb = g.tpb;
        tp.off_epi = g.off_epi;
        tp.off_act = g.off_act;
        tp.cin = n->d.in_planes;
        tp.n_convs = 2 * n->d.n_blocks;
        tp.n_boards = n_boards;
        tp.bpw = 1;
        tp.rcells = g.rcells;
        tp.zcell = g.zcell;
        tp.conv_w = n->conv_w;
        tp.epi = n->epi;
        tp.skip_w = n->skip_w;
        memcpy(tp.in_scale, n->in_affine, 32);
        memcpy(tp.in_shift, n->in_affine + 8, 32);
        tp.obs = obs;
        tp.xout = n->xout;
        tp.xout_lo = n->xout_lo;
        tp.xout_c = n->xc;
        const int grid = (n_boards + 3) / 4;
        const NetDispatch dp = net_dispatch(n, n_boards);
        const bool x3c = dp.x3c; // small batch: one board per workgroup (az_tower_x3c.h)
        // ... and with a single output tile (connect_four: 7 + 1 outputs) that kernel also runs fc1 + softmax + tanh for its board:
        // at <= 512 boards the head kernel is 7 us of a 49 us tick (profiles/r3_small_generation_kernel_stats.csv).  (The same inside
        // az_tower_x3b_kernel, measured: bit-identical and SLOWER - 1146 vs 1182 games/s in a same-box A/B: at one workgroup per CU
        // the head phase of each of the four rounds, ~5 us, has nothing to hide behind.)
        const bool fused_head = dp.x3c_head; // (eight chains x HMAX k-steps; a chain steps 4 columns: az_tower_x3c.h)
        tp.fc_w = fused_head ? n->fc_w : nullptr;
        tp.fc_w_lo = n->fc_w_lo;
        tp.fc_b = n->fc_b;
        tp.priors = priors;
        tp.values = values;
        tp.A = n->d.num_actions;
        tp.fc_ksteps = hp.ksteps;
        hipError_t s;
        if (x3c) s = az_launch_tower_x3c(n->d.device, n_boards > AZ_X3C_ONE_PER_WG ? 2 : 1, tp, n_boards, st);
        else if (dp.x3d) { // packed column tiles (az_tower_x3d.h)
            tp.conv_w = n->conv_w_d;
            tp.xd_nb = n->xd_nb, tp.xd_R = n->xd_R, tp.xd_rs = n->xd_rs;
            tp.xd_pos = n->xd_pos, tp.xd_sdst = n->xd_sdst;
            s = az_launch_tower_x3d(n->d.device, n->x3d, tp, (n_boards + n->xd_nb - 1) / n->xd_nb, st);
        } else if (n->x3b) s = az_launch_tower_x3b(n->d.device, tp, grid, st);
        else s = az_launch_tower_x3(n->d.device, g.nt, g.rp1, n->r3, tp, grid, g.lds, st);
        if (s != hipSuccess) {
            n->err = std::string("f16x3 tower launch: ") + hipGetErrorString(s);
            return AZ_E_HIP;
        }
        if (!fused_head) NCHK(n, az_launch_head(n->d.device, true, hp, n_boards, n->lds_head, n->logits, st));
        return AZ_OK;
    }
    const TowerGeom g = choose_geom(n, n_boards);
    TowerParams tp;
    tp.H = n->d.rows;
    tp.W = n->d.cols;
    tp.HW = tp.H * tp.W;
    tp.cells = g.cells;
    tp.rs = g.rs;
    tp.tpb = g.tpb;
    tp.off_epi = g.off_epi;
    tp.cin = n->d.in_planes;
    tp.n_convs = 2 * n->d.n_blocks;
    tp.n_boards = n_boards;
    tp.bpw = g.bpw;
    tp.rcells = g.rcells;
    tp.zcell = g.zcell;
    tp.off_act = g.off_act;
    tp.conv_w = n->conv_w;
    tp.epi = n->epi;
    tp.skip_w = n->skip_w;
    memcpy(tp.in_scale, n->in_affine, 32);
    memcpy(tp.in_shift, n->in_affine + 8, 32);
    tp.obs = obs;
    tp.xout = n->xout;
    tp.xout_lo = nullptr;
    tp.xout_c = n->xc;
    tp.fc_w = nullptr;
    int per_wg = g.waves * g.bpw, grid = (n_boards + per_wg - 1) / per_wg;
    hipError_t s;
    // small batch of a row-pair board with <= 50 filters: a board per four-wave workgroup (az_tower_f16c.h; same bits)
    const TowerGeom gc = tower_geom(1, 1, n->d.rows, n->d.cols);
    if (net_dispatch(n, n_boards).f16c) {
        // up to a board per CU: 32 KiB weight chunks (half the barriers); above: 16 KiB chunks, so that two workgroups fit a CU's LDS
        const int ck = n_boards <= AZ_X3C_ONE_PER_WG ? 8 : 4;
        tp.cells = gc.cells, tp.rs = gc.rs, tp.tpb = gc.tpb;
        tp.off_epi = (ck == 8 ? 4 : 3) * ck * 4096; // a ring of four 32-KiB / three 16-KiB weight buffers (147 / 65 KB with the planes)
        tp.off_act = tp.off_epi + 2048 + 8 * 64 * 8;
        tp.bpw = 1, tp.rcells = gc.rcells, tp.zcell = gc.zcell;
        s = az_launch_tower_f16c(n->d.device, ck, tp, n_boards, tp.off_act + N_OCT * gc.rcells * OCT_B, st);
    } else
        s = az_launch_tower_f16(n->d.device, g.nt, g.ck, g.waves, n->r3, tp, grid, g.lds, st);
    if (s != hipSuccess) {
        n->err = std::string("tower launch: ") + hipGetErrorString(s);
        return AZ_E_HIP;
    }
    NCHK(n, az_launch_head(n->d.device, false, hp, n_boards, n->lds_head, n->logits, st));
    return AZ_OK;
}

extern "C" int az_net_read_tower(az_net *n, float *out, int32_t n_boards) {
    if (!n || !out || n_boards < 1 || n_boards > n->max_boards) return AZ_E_INVALID;
    NCHK(n, hipSetDevice(n->d.device));
    NCHK(n, hipDeviceSynchronize());
    // out: [n_boards][H*W][AZ_NET_XOUT_C] whatever the stride on the device (channels past it read as zero)
    const size_t cells = (size_t)n_boards * n->d.rows * n->d.cols, cnt = cells * n->xc;
    const int nc = n->xc < AZ_NET_XOUT_C ? n->xc : AZ_NET_XOUT_C;
    std::vector<_Float16> h(cnt);
    NCHK(n, hipMemcpy(h.data(), n->xout, cnt * 2, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < cells * AZ_NET_XOUT_C; i++) out[i] = 0.f;
    for (size_t c = 0; c < cells; c++)
        for (int ch = 0; ch < nc; ch++) out[c * AZ_NET_XOUT_C + ch] = (float)h[c * n->xc + ch];
    if (n->precision == AZ_NET_PREC_F16X3) { // hi + lo / 2048
        NCHK(n, hipMemcpy(h.data(), n->xout_lo, cnt * 2, hipMemcpyDeviceToHost));
        for (size_t c = 0; c < cells; c++)
            for (int ch = 0; ch < nc; ch++) out[c * AZ_NET_XOUT_C + ch] += (float)h[c * n->xc + ch] * (1.0f / 2048.0f);
    }
    return AZ_OK;
}
