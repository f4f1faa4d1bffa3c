// az_head.h — fc1 + softmax + tanh of Net.forward (network.py:61-64): az_head_kernel (small action spaces),
// az_head_gemm_kernel + az_head_softmax_kernel (breakthrough's 433 / 769 outputs).
#pragma once
#include "az_head_params.h"

// The A operand of fc1 is 16 boards x 32 k per k-step, and the MFMA wants board l15's octet q in lane 16 q + l15.  Loaded that way,
// the four neighbouring lanes of a load touch four different boards' rows: the texture addresser takes the 64 lanes as 64 separate
// 16-byte accesses (rocprof, 6x6 logits kernel: 40 accesses per vector-memory instruction, the addresser busy 68 % of the kernel).
// So lane L LOADS octet L & 3 of board L >> 2 - four neighbouring lanes = 64 contiguous bytes, 16 accesses per instruction - and
// one ds_bpermute per dword then hands lane 16 q + r the registers of lane 4 r + q (no LDS memory involved).
__device__ __forceinline__ half8 frag_from_rows(half8 v, int lane) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const int src = (4 * (lane & 15) + (lane >> 4)) * 4;
    i32x4 u = __builtin_bit_cast(i32x4, v);
#pragma unroll
    for (int i = 0; i < 4; i++) u[i] = __builtin_amdgcn_ds_bpermute(src, u[i]);
    return __builtin_bit_cast(half8, u);
}

// ------------------------------------------------------------------------------------------------
// fc1 + softmax + tanh (network.py:61-64).  One workgroup = 16 boards; the K = HW*64 reduction is split
// over the 4 waves (k-step ks goes to wave ks & 3), partial tiles are summed through LDS.
// X3: split-fp16 operands (see az_tower_x3_kernel): three MFMAs per product, result = acc + acc2 / 2048.
template <bool X3> __global__ __launch_bounds__(HEAD_NW * 64) void az_head_kernel(HeadParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float *part = (float *)lds;                               // [HEAD_NW waves][OTG][64 lanes][4]
    float *logits = (float *)(lds + HEAD_NW * OTG * 64 * 16); // [16][n_ot*16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b0 = blockIdx.x * 16;
    const int K = p.K, NP = p.n_ot * 16;
    int row = b0 + (lane >> 2); // (load mapping of frag_from_rows)
    if (row >= p.n_boards) row = p.n_boards - 1; // clamp: computed, never stored
    const _Float16 *xrow = p.x + (size_t)row * K + 8 * (lane & 3);
    const _Float16 *xrow_lo = X3 ? p.x_lo + (size_t)row * K + 8 * (lane & 3) : nullptr;
    for (int og = 0; og < p.n_ot; og += OTG) {
        f32x4 acc[OTG], acc2[X3 ? OTG : 1];
#pragma unroll
        for (int o = 0; o < OTG; o++) acc[o] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int o = 0; o < (X3 ? OTG : 1); o++) acc2[o] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (p.n_ot == 1) { // small action space (connect_four): one output tile -> a pure chain of load, load, MFMA per
                           // k-step; unrolled so that the loads of several k-steps are in flight together
#pragma unroll 8
            for (int ks = wave; ks < p.ksteps; ks += HEAD_NW) {
                half8 a = frag_from_rows(*(const half8 *)(xrow + 32 * ks), lane);
                half8 w = *(const half8 *)(p.fc_w + ((size_t)ks * 64 + lane) * 8);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, w, acc[0], 0, 0, 0);
                if constexpr (X3) {
                    half8 al = frag_from_rows(*(const half8 *)(xrow_lo + 32 * ks), lane);
                    half8 wl = *(const half8 *)(p.fc_w_lo + ((size_t)ks * 64 + lane) * 8);
                    acc2[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, wl, acc2[0], 0, 0, 0);
                    acc2[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, w, acc2[0], 0, 0, 0);
                }
            }
        } else
        for (int ks = wave; ks < p.ksteps; ks += HEAD_NW) {
            half8 a = frag_from_rows(*(const half8 *)(xrow + 32 * ks), lane);
            half8 al;
            if constexpr (X3) al = frag_from_rows(*(const half8 *)(xrow_lo + 32 * ks), lane);
#pragma unroll
            for (int o = 0; o < OTG; o++)
                if (og + o < p.n_ot) {
                    const size_t wi = (((size_t)(og + o) * p.ksteps + ks) * 64 + lane) * 8;
                    half8 w = *(const half8 *)(p.fc_w + wi);
                    acc[o] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, w, acc[o], 0, 0, 0);
                    if constexpr (X3) {
                        half8 wl = *(const half8 *)(p.fc_w_lo + wi);
                        acc2[o] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, wl, acc2[o], 0, 0, 0);
                        acc2[o] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, w, acc2[o], 0, 0, 0);
                    }
                }
        }
#pragma unroll
        for (int o = 0; o < OTG; o++) {
            if constexpr (X3) acc[o] = acc[o] + acc2[o] * (1.0f / 2048.0f);
            *(f32x4 *)(part + ((wave * OTG + o) * 64 + lane) * 4) = acc[o];
        }
        __syncthreads();
        // the threads sum the HEAD_NW partials of OTG*64 float4 slots
        for (int s = tid; s < OTG * 64; s += HEAD_NW * 64) {
            int o = s >> 6, ln = s & 63;
            if (og + o >= p.n_ot) continue;
            f32x4 v = *(f32x4 *)(part + ((0 * OTG + o) * 64 + ln) * 4);
#pragma unroll
            for (int w = 1; w < HEAD_NW; w++) v += *(f32x4 *)(part + ((w * OTG + o) * 64 + ln) * 4);
            int col = 16 * (og + o) + (ln & 15);
            float bias = p.fc_b[col];
#pragma unroll
            for (int r = 0; r < 4; r++) logits[((ln >> 4) * 4 + r) * NP + col] = v[r] + bias; // D: row = 4q+r, col = l15
        }
        __syncthreads();
    }
    // softmax over the first A logits, tanh of logit A: 16 lanes per board
    const int brd = (tid >> 4) & 15, sub = tid & 15; // (threads 256.. repeat the work of 0..255 and store nothing)
    const float *lg = logits + brd * NP;
    float mx = -INFINITY;
    for (int o = sub; o < p.A; o += 16) mx = fmaxf(mx, lg[o]);
#pragma unroll
    for (int off = 8; off; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 16));
    float sum = 0.f;
    for (int o = sub; o < p.A; o += 16) sum += X3 ? expf(lg[o] - mx) : __expf(lg[o] - mx);
#pragma unroll
    for (int off = 8; off; off >>= 1) sum += __shfl_xor(sum, off, 16);
    if (tid < 256 && b0 + brd < p.n_boards) {
        float *out = p.priors + (size_t)(b0 + brd) * p.A;
        if (X3) {
            for (int o = sub; o < p.A; o += 16) out[o] = expf(lg[o] - mx) / sum;
        } else {
            float inv = 1.f / sum;
            for (int o = sub; o < p.A; o += 16) out[o] = __expf(lg[o] - mx) * inv;
        }
        if (sub == 0) p.values[b0 + brd] = tanhf(lg[p.A]);
    }
}

// Large action spaces (breakthrough: 433 / 769 outputs = 28 / 49 output tiles): fc1 is a real GEMM there,
//   logits[board][o] = sum_k x[board][k] * Wfc[o][k],   M = boards, N = n_ot * 16, K = H*W*52 (az_net_create),
// tiled with BOTH operands staged through LDS.  A workgroup owns HG_BT x 16 = 128 boards x up to HG_OT x 16 = 128 outputs over HALF
// of K (HG_KSPLIT = 2: 2048 boards x 769 outputs are only 112 such tiles for 256 CUs; az_head_softmax_kernel adds the two partial
// sums, always in the same order, so a board's result does not depend on the batch).  The output tiles are dealt evenly to the
// groups (49 = 7 x 7, 28 = 4 x 7: no padding tile is multiplied).  Eight computing waves (wm, wn) = 32 boards x 4 or 3 output tiles
// over the workgroup's K: per k-step 32 KiB come in for 128 x 128 x 32 products, 12 fragment reads feed 24 (21) MFMAs; HG_LOADERS
// more waves do nothing but issue the LDS-DMA (a computing wave that issues its own share spends MFMA slots on it: 51 -> 45 us).
//   * a weight fragment is one contiguous KiB of the packed fc stream;
//   * an A fragment is 16 boards x 64 bytes of the tower output.  Lane 4r + j fetches octet j ^ 2 (r >> 3) of board r (four
//     neighbouring lanes = 64 contiguous bytes, see frag_from_rows), the DMA drops it at byte 16 (4r + j) of the KiB, and the MFMA
//     lane (q, r) reads byte 16 (4r + (q ^ 2 (r >> 3))).  ds_read_b128 serves the lanes in the groups {0-3, 12-15, 20-27}, {4-11,
//     16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS): a group holds every board r once, with octet q0 + [4 <= r < 12] or q0 + 1 -
//     [4 <= r < 12], and its sixteen 16-byte bank groups 4 (r & 3) + (q ^ 2 (r >> 3)) are all different - no bank conflict, no
//     ds_bpermute;
//   * HG_RING k-step slots; the loaders wait with counted vmcnt (a __syncthreads() would drain the k-steps in flight), one bare
//     barrier per k-step hands a slot from the loaders to the computing waves and the slot behind it back.
// Board tiles past n_boards are fetched clamped and never multiplied (wave-uniform), never stored.
// History: round 1 read the fc weights once per 16 boards (45 / 82 us for 4096 6x6 / 2048 8x8 boards); rounds 2-3 ran 64 boards x 64
// outputs per workgroup with the A fragments straight from L2 (44 / 85 us: 832 MB through the L2s per 2048 8x8 boards, and a branch
// and a wait around every MFMA of the 3-tile waves would have been the next problem); this kernel: 32 / 43 us.
#define HG_BT 8
#define HG_OT 8
#define HG_KSPLIT 2
#define HG_RING 4
#ifndef HG_LOADERS
#define HG_LOADERS 4 // waves that only issue the LDS-DMA (2 measured the same)
#endif
template <bool X3> __global__ __launch_bounds__((8 + HG_LOADERS) * 64) void az_head_gemm_kernel(HeadParams p, float *__restrict__ logits_g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int NPART = X3 ? 2 : 1;
    constexpr int PIECES = (HG_BT + HG_OT) * NPART; // KiB fragments per k-step: [A part][board tile], [W part][output tile]
    constexpr int SLOT_B = PIECES * 1024;
    constexpr int NLOAD = HG_LOADERS;                  // waves that issue pieces
    constexpr int PER = PIECES / NLOAD;               // per loading wave and k-step
    static_assert((NLOAD == 2 || NLOAD == 4) && PIECES % NLOAD == 0 && HG_BT == 8 && HG_OT == 8 && (HG_RING - 2) * PER < 64, "piece f = NLOAD i + loader; the counted waits fit vmcnt");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), q = lane >> 4, l15 = lane & 15;
    // wave (wm, wn): board tiles 2 wm, 2 wm + 1 x output tiles 4 wn .. 4 wn + 3 of the group; the two waves of a SIMD (w, w + 4) take
    // one half of the group's tiles each (a group of 7: 4 + 3 on every SIMD)
    const int wm = (wave & 7) >> 1, wn = (wave ^ (wave >> 2)) & 1;
    // XCD-aware order.  Workgroup L runs on XCD L % 8 (each XCD has its own L2): the workgroups of one board tile get consecutive
    // slots of ONE XCD, so those boards' rows come in from the Infinity Cache once and are shared in that L2
    const int n_cg = (p.n_ot + HG_OT - 1) / HG_OT, ot_per = (p.n_ot + n_cg - 1) / n_cg, per_bt = n_cg * HG_KSPLIT;
    const int lin = blockIdx.x, xcd = lin & 7, j = lin >> 3;
    const int bt = (j / per_bt) * 8 + xcd;
    if (bt * 16 * HG_BT >= p.n_boards) return; // (whole workgroup: before any barrier)
    const int og = ((j % per_bt) / HG_KSPLIT) * ot_per, kh = (j % per_bt) % HG_KSPLIT;
    const int n_og = p.n_ot - og < ot_per ? p.n_ot - og : ot_per; // output tiles of this group: 49 = 7 x 7, 28 = 4 x 7 (no padding tile)
    const int K = p.K, NP = p.n_ot * 16;
    const int ks_per = (p.ksteps + HG_KSPLIT - 1) / HG_KSPLIT, ks0 = kh * ks_per;
    const int nk = (p.ksteps - ks0 < ks_per ? p.ksteps - ks0 : ks_per);
    const bool loader = wave >= 8;
    // a loading wave's PER pieces of a k-step: base address at k-step 0 and halves per k-step
    const _Float16 *src0[PER];
    int kstride[PER], dst[PER];
    if (loader) {
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int f = NLOAD * i + (wave & (NLOAD - 1));
            dst[i] = f * 1024;
            if (f < HG_BT * NPART) {
                const int r = lane >> 2;
                int row = (bt * HG_BT + f % HG_BT) * 16 + r;
                if (row >= p.n_boards) row = p.n_boards - 1;
                src0[i] = (f / HG_BT ? p.x_lo : p.x) + (size_t)row * K + 8 * ((lane & 3) ^ ((r >> 3) << 1));
                kstride[i] = 32;
            } else {
                const int g = f - HG_BT * NPART;
                int ot = og + g % HG_OT;
                ot = ot < p.n_ot ? ot : p.n_ot - 1; // (a tile past the group: a valid fragment, never multiplied)
                src0[i] = (g / HG_OT ? p.fc_w_lo : p.fc_w) + ((size_t)ot * p.ksteps * 64 + lane) * 8;
                kstride[i] = 512;
            }
        }
    }
    auto issue = [&](int ks, int slot) {
#pragma unroll
        for (int i = 0; i < PER; i++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src0[i] + (size_t)ks * kstride[i]),
                                             (__attribute__((address_space(3))) void *)(lds + slot * SLOT_B + dst[i]), 16, 0, 0);
    };
    // The hand-over, per k-step i: barrier i tells the computing waves that k-step i + 1 is in LDS and the loaders that everybody has
    // read k-step i (a computing wave reads k-step i + 1 into its second register set right behind barrier i, then multiplies k-step
    // i: the reads' latency hides behind the MFMAs).  One barrier before the loop hands over k-step 0.
    if (loader) {
#pragma unroll
        for (int s = 0; s < HG_RING - 1; s++)
            if (s < nk) issue(ks0 + s, s);
        for (int i = -1; i < nk; i++) {
            if (i + 1 < nk) { // k-step i + 1 has landed once at most the pieces of the (up to RING - 2) younger k-steps are outstanding
                const int younger = nk - 2 - i < HG_RING - 2 ? nk - 2 - i : HG_RING - 2;
                static_for<HG_RING - 1>([&](auto y_c) { // (a literal operand per possible count)
                    constexpr int y = decltype(y_c)::value;
                    if (younger == y) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(y * PER) : "memory");
                });
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (i + HG_RING < nk) issue(ks0 + i + HG_RING, (i + HG_RING) % HG_RING); // into the slot of k-step i
        }
        return;
    }
    // board tiles / output tiles this wave multiplies (wave-uniform): both board tiles and 4 or 3 output tiles in all but the last
    // board tile of a ragged batch and the last group of an odd split - those take the predicated loop
    const int n_m = (bt * HG_BT + 2 * wm + 1) * 16 < p.n_boards ? 2 : (bt * HG_BT + 2 * wm) * 16 < p.n_boards ? 1 : 0;
    const int n_o = n_og - 4 * wn < 0 ? 0 : n_og - 4 * wn > 4 ? 4 : n_og - 4 * wn;
    // the biases come in HERE (vmcnt counts stores too: a bias load between the output tiles' stores waits for the stores before it);
    // they ride on the first half of K
    float bias_o[4];
#pragma unroll
    for (int o = 0; o < 4; o++) bias_o[o] = kh == 0 && o < n_o ? p.fc_b[16 * (og + 4 * wn + o) + l15] : 0.f;
    f32x4 acc[2][4], acc2[X3 ? 2 : 1][X3 ? 4 : 1];
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int o = 0; o < 4; o++) {
            acc[m][o] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (X3) acc2[m][o] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    const int a_off = (4 * l15 + (q ^ ((l15 >> 3) << 1))) * 16 + 2 * wm * 1024;
    const int w_off = (HG_BT * NPART + 4 * wn) * 1024 + lane * 16;
    // NO = 4, 3: straight-line k-step for two board tiles x NO output tiles; NO = 0: every tile behind a (scalar) branch
    auto kloop = [&](auto no_c) {
        constexpr int NO = decltype(no_c)::value, NOT = NO ? NO : 4;
        half8 af[2], afl[X3 ? 2 : 1], w[4], wl[X3 ? 4 : 1];
        auto read = [&](int i) {
            const unsigned char *sb = lds + (i % HG_RING) * SLOT_B;
#pragma unroll
            for (int m = 0; m < 2; m++) {
                af[m] = *(const half8 *)(sb + a_off + m * 1024);
                if constexpr (X3) afl[m] = *(const half8 *)(sb + a_off + (HG_BT + m) * 1024);
            }
#pragma unroll
            for (int o = 0; o < NOT; o++) {
                w[o] = *(const half8 *)(sb + w_off + o * 1024);
                if constexpr (X3) wl[o] = *(const half8 *)(sb + w_off + (HG_OT + o) * 1024);
            }
        };
        auto step = [&](int i) {
            read(i);
            // three sweeps over the wave's tiles: the two MFMAs into acc2[m][o] are a sweep apart (back to back, the second waits out
            // the first one's latency)
#pragma unroll
            for (int pass = 0; pass < (X3 ? 3 : 1); pass++)
#pragma unroll
                for (int o = 0; o < NOT; o++)
#pragma unroll
                    for (int m = 0; m < 2; m++)
                        if (NO || (o < n_o && m < n_m)) {
                            if (pass == 0) acc[m][o] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[m], w[o], acc[m][o], 0, 0, 0);
                            if constexpr (X3) {
                                if (pass == 1) acc2[m][o] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[m], wl[o], acc2[m][o], 0, 0, 0);
                                if (pass == 2) acc2[m][o] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afl[m], w[o], acc2[m][o], 0, 0, 0);
                            }
                        }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier(); // k-step i + 1 is in LDS; the loaders may refill the slot of k-step i
            asm volatile("" ::: "memory");
        };
        __builtin_amdgcn_s_barrier(); // k-step 0 is in LDS
        asm volatile("" ::: "memory");
        for (int i = 0; i < nk; i++) step(i);
    };
    if (n_m == 2 && n_o == 4) kloop(std::integral_constant<int, 4>{});
    else if (n_m == 2 && n_o == 3) kloop(std::integral_constant<int, 3>{});
    else kloop(std::integral_constant<int, 0>{});
    // D: row = 4q + r -> board, col = l15 -> output 16 (og + 4 wn + o) + l15
    float *out = logits_g + (size_t)kh * p.n_boards * NP;
#pragma unroll
    for (int o = 0; o < 4; o++) {
        if (o >= n_o) continue;
        const int col = 16 * (og + 4 * wn + o) + l15;
        const float bias = bias_o[o];
#pragma unroll
        for (int m = 0; m < 2; m++) {
            f32x4 v = acc[m][o];
            if constexpr (X3) v = v + acc2[m][o] * (1.0f / 2048.0f);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int b = (bt * HG_BT + 2 * wm + m) * 16 + 4 * q + r;
                if (b < p.n_boards) out[(size_t)b * NP + col] = v[r] + bias;
            }
        }
    }
}
// softmax over the first A logits, tanh of logit A: one WAVE per board, the board's logits held in registers
#define HEAD_SM_MAX 13 // ceil((12 * 64 + 1) / 64): A <= 768 (boards of <= 64 cells)
// NSPLIT: partial sums over K to add (az_head_gemm_kernel: HG_KSPLIT, buffers n_boards * NP floats apart)
template <bool X3, int NSPLIT> __global__ __launch_bounds__(256) void az_head_softmax_kernel(HeadParams p, const float *__restrict__ logits_g) {
    const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6), NP = p.n_ot * 16;
    if (b >= p.n_boards) return;
    const float *lg = logits_g + (size_t)b * NP;
    const size_t part = (size_t)p.n_boards * NP;
    auto logit = [&](int o) {
        float v = lg[o];
#pragma unroll
        for (int s = 1; s < NSPLIT; s++) v += lg[s * part + o];
        return v;
    };
    float v[HEAD_SM_MAX];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < HEAD_SM_MAX; i++) {
        const int o = i * 64 + lane;
        v[i] = o < p.A ? logit(o) : -INFINITY;
        mx = fmaxf(mx, v[i]);
    }
#pragma unroll
    for (int off = 32; off; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < HEAD_SM_MAX; i++) {
        v[i] = i * 64 + lane < p.A ? (X3 ? expf(v[i] - mx) : __expf(v[i] - mx)) : 0.f;
        sum += v[i];
    }
#pragma unroll
    for (int off = 32; off; off >>= 1) sum += __shfl_xor(sum, off);
    float *out = p.priors + (size_t)b * p.A;
    const float inv = 1.f / sum;
#pragma unroll
    for (int i = 0; i < HEAD_SM_MAX; i++) {
        const int o = i * 64 + lane;
        if (o < p.A) out[o] = X3 ? v[i] / sum : v[i] * inv;
    }
    if (lane == 0) p.values[b] = tanhf(logit(p.A));
}
