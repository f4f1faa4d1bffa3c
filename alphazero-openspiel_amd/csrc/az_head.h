// az_head.h — fc1 + softmax + tanh of Net.forward (network.py:61-64): az_head_kernel (small action spaces),
// az_head_logits_kernel + az_head_softmax_kernel (breakthrough's 433 / 769 outputs).
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
    const int K = p.HW * AZ_NET_XOUT_C, NP = p.n_ot * 16;
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
//   logits[board][o] = sum_k x[board][k] * Wfc[o][k],   M = boards, N = n_ot * 16, K = H*W*64.
// Round 1's kernel (16 boards x 8 output tiles per workgroup, K split over the waves, every wave pulling its own weight
// fragments from L2) re-read the 2-6 MB of fc weights once per 16 boards: 45 us (6x6, 4096 boards) / 82 us (8x8, 2048
// boards) at 7 % of the matrix peak - 18 % of those configurations' GPU time (profiles/r2_c3_kernel_stats.csv).
// Now: a workgroup = 4 waves = 128 boards x HEAD_OTG output tiles; every wave owns 2 x 16 boards over the WHOLE K (no
// cross-wave reduction; a weight fragment read from LDS feeds two MFMAs); the weight fragments of a chunk of HEAD_CK k-steps are brought into LDS ONCE per workgroup by LDS-DMA
// (a fragment is one contiguous KiB = one wave-instruction), double buffered; A fragments come straight from the tower
// output (each wave reads only its own boards) and are prefetched a chunk ahead.  L2 traffic for the weights drops 4x.
// HEAD_MT: board tiles (x16 boards) per wave - every weight fragment read from LDS feeds HEAD_MT MFMAs
template <bool X3, int HEAD_MT> __global__ __launch_bounds__(256) void az_head_logits_kernel(HeadParams p, float *__restrict__ logits_g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int CK = X3 ? 2 : 4;                       // k-steps per chunk: 16 KiB of weight fragments either way
    constexpr int NPART = X3 ? 2 : 1;
    constexpr int FRAGS = CK * HEAD_OTG * NPART;         // KiB fragments per chunk: [part][ksl][o]
    constexpr int CHUNK_B = FRAGS * 1024;
    constexpr int PER = FRAGS / 4 + CK * NPART * HEAD_MT; // vector-memory operations one wave issues per chunk
    static_assert(FRAGS % 4 == 0 && (HEAD_RING - 2) * PER < 64, "pieces split evenly over the 4 waves; the counted waits fit vmcnt");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, l15 = lane & 15;
    // XCD-aware tile order.  Workgroup L runs on XCD L % 8 (each XCD has its own L2): the column groups of one board tile get
    // CONSECUTIVE slots of ONE XCD, so the tower output of those boards comes in from the Infinity Cache / HBM once and is
    // re-read from that XCD's L2 by the other column groups (a plain 2-D grid re-fetched it once per column group).
    const int n_cg = (p.n_ot + HEAD_OTG - 1) / HEAD_OTG;
    const int lin = blockIdx.x, xcd = lin & 7, j = lin >> 3;
    const int bt = (j / n_cg) * 8 + xcd;
    if (bt * 64 * HEAD_MT >= p.n_boards) return; // (whole workgroup: before any barrier)
    const int b0 = (bt * 4 + wave) * 16 * HEAD_MT, og = (j % n_cg) * HEAD_OTG;
    const int K = p.HW * AZ_NET_XOUT_C, NP = p.n_ot * 16;
    const int n_chunks = (p.ksteps + CK - 1) / CK;
    const _Float16 *xrow[HEAD_MT], *xrow_lo[HEAD_MT];
#pragma unroll
    for (int m = 0; m < HEAD_MT; m++) {
        int row = b0 + 16 * m + (lane >> 2); // (load mapping of frag_from_rows)
        if (row >= p.n_boards) row = p.n_boards - 1; // clamp: computed, never stored
        xrow[m] = p.x + (size_t)row * K + 8 * (lane & 3);
        xrow_lo[m] = X3 ? p.x_lo + (size_t)row * K + 8 * (lane & 3) : nullptr;
    }
    half8 a[HEAD_RING][CK][HEAD_MT], al[HEAD_RING][X3 ? CK : 1][X3 ? HEAD_MT : 1];
    // chunk c -> LDS slot `slot` (compile-time) + the A fragments of its k-steps.  Out-of-range tiles / k-steps re-fetch a valid
    // fragment (their products are never stored / never accumulated).
    auto issue_chunk = [&](int c, auto slot_c) {
        constexpr int slot = decltype(slot_c)::value;
#pragma unroll
        for (int i = 0; i < FRAGS / 4; i++) {
            const int f = i * 4 + wave;
            const int part = f / (CK * HEAD_OTG), ksl = (f / HEAD_OTG) % CK, o = f % HEAD_OTG;
            int ot = og + o, ks = c * CK + ksl;
            ot = ot < p.n_ot ? ot : p.n_ot - 1;
            ks = ks < p.ksteps ? ks : p.ksteps - 1;
            const _Float16 *src = (part ? p.fc_w_lo : p.fc_w) + (((size_t)ot * p.ksteps + ks) * 64 + lane) * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(lds + slot * CHUNK_B + f * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int ksl = 0; ksl < CK; ksl++) {
            int ks = c * CK + ksl;
            ks = ks < p.ksteps ? ks : p.ksteps - 1;
#pragma unroll
            for (int m = 0; m < HEAD_MT; m++) {
                // asm, not a C++ load: the counted s_waitcnt below relies on the ISSUE ORDER of every vector-memory operation
                // (a compiler-scheduled load could be sunk towards its use and shift the count)
                {
                    half8 &dst = a[slot][ksl][m];
                    const _Float16 *src = xrow[m] + 32 * ks;
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(src) : "memory");
                }
                if constexpr (X3) {
                    half8 &dst = al[slot][ksl][m];
                    const _Float16 *src = xrow_lo[m] + 32 * ks;
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(src) : "memory");
                }
            }
        }
    };
    // The biases come in HERE.  vmcnt counts stores too on this chip: a bias load between the output tiles' stores waited for the
    // stores before it - four store round trips in a row, 65 % of the kernel (clock64 probes, 6x6).
    float bias_o[HEAD_OTG];
#pragma unroll
    for (int o = 0; o < HEAD_OTG; o++) bias_o[o] = p.fc_b[16 * (og + o < p.n_ot ? og + o : p.n_ot - 1) + l15];
    f32x4 acc[HEAD_MT][HEAD_OTG], acc2[X3 ? HEAD_MT : 1][X3 ? HEAD_OTG : 1];
#pragma unroll
    for (int m = 0; m < HEAD_MT; m++)
#pragma unroll
        for (int o = 0; o < HEAD_OTG; o++) {
            acc[m][o] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (X3) acc2[m][o] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    static_for<HEAD_RING - 1>([&](auto s_c) {
        if (decltype(s_c)::value < n_chunks) issue_chunk(decltype(s_c)::value, s_c);
    });
    for (int c0 = 0; c0 < n_chunks; c0 += HEAD_RING) {
        static_for<HEAD_RING>([&](auto s_c) {
            constexpr int slot = decltype(s_c)::value;
            const int c = c0 + slot;
            if (c < n_chunks) {
                // chunk c has landed once at most the operations of the (up to RING - 2) younger chunks are outstanding
                const int younger = n_chunks - 1 - c < HEAD_RING - 2 ? n_chunks - 1 - c : HEAD_RING - 2;
                static_for<HEAD_RING - 1>([&](auto y_c) { // (a literal operand per possible count)
                    constexpr int y = decltype(y_c)::value;
                    if (younger == y) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(y * PER) : "memory");
                });
                // a BARE barrier: __syncthreads() carries a fence that drains vmcnt to 0 and with it the chunks in flight
                __builtin_amdgcn_s_barrier(); // everybody's pieces of chunk c are in LDS, and the slot of chunk c - 1 is free again
                asm volatile("" ::: "memory");
                if (c + HEAD_RING - 1 < n_chunks) issue_chunk(c + HEAD_RING - 1, std::integral_constant<int, (slot + HEAD_RING - 1) % HEAD_RING>{});
                const unsigned char *wb = lds + slot * CHUNK_B + lane * 16;
                // One straight-line block per chunk: the A fragments' lane exchange and ALL the chunk's weight-fragment reads go out first,
                // the MFMAs follow behind counted waits (a branch per k-step left each k-step waiting out two LDS round trips).  A k-step
                // past the end (odd H*W only) multiplies a zero A fragment.
                half8 af[CK][HEAD_MT], afl[X3 ? CK : 1][X3 ? HEAD_MT : 1];
                half8 w[CK][HEAD_OTG], wl[X3 ? CK : 1][X3 ? HEAD_OTG : 1];
                const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int ksl = 0; ksl < CK; ksl++) {
                    const bool live = c * CK + ksl < p.ksteps;
#pragma unroll
                    for (int m = 0; m < HEAD_MT; m++) {
                        af[ksl][m] = frag_from_rows(live ? a[slot][ksl][m] : zero8, lane);
                        if constexpr (X3) afl[ksl][m] = frag_from_rows(live ? al[slot][ksl][m] : zero8, lane);
                    }
                }
#pragma unroll
                for (int ksl = 0; ksl < CK; ksl++)
#pragma unroll
                    for (int o = 0; o < HEAD_OTG; o++) {
                        w[ksl][o] = *(const half8 *)(wb + (ksl * HEAD_OTG + o) * 1024);
                        if constexpr (X3) wl[ksl][o] = *(const half8 *)(wb + ((CK + ksl) * HEAD_OTG + o) * 1024);
                    }
#pragma unroll
                for (int ksl = 0; ksl < CK; ksl++)
#pragma unroll
                    for (int o = 0; o < HEAD_OTG; o++)
#pragma unroll
                        for (int m = 0; m < HEAD_MT; m++) {
                            acc[m][o] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[ksl][m], w[ksl][o], acc[m][o], 0, 0, 0);
                            if constexpr (X3) {
                                acc2[m][o] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[ksl][m], wl[ksl][o], acc2[m][o], 0, 0, 0);
                                acc2[m][o] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afl[ksl][m], w[ksl][o], acc2[m][o], 0, 0, 0);
                            }
                        }
            }
        });
    }
    // D: row = 4q + r -> board b0 + 16 m + 4q + r, col = l15 -> output 16 (og + o) + l15
#pragma unroll
    for (int o = 0; o < HEAD_OTG; o++) {
        if (og + o >= p.n_ot) continue;
        const int col = 16 * (og + o) + l15;
        const float bias = bias_o[o];
#pragma unroll
        for (int m = 0; m < HEAD_MT; m++) {
            f32x4 v = acc[m][o];
            if constexpr (X3) v = v + acc2[m][o] * (1.0f / 2048.0f);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int b = b0 + 16 * m + 4 * q + r;
                if (b < p.n_boards) logits_g[(size_t)b * NP + col] = v[r] + bias;
            }
        }
    }
}
// softmax over the first A logits, tanh of logit A: one WAVE per board, the board's logits held in registers
#define HEAD_SM_MAX 13 // ceil((12 * 64 + 1) / 64): A <= 768 (boards of <= 64 cells)
template <bool X3> __global__ __launch_bounds__(256) void az_head_softmax_kernel(HeadParams p, const float *__restrict__ logits_g) {
    const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6), NP = p.n_ot * 16;
    if (b >= p.n_boards) return;
    const float *lg = logits_g + (size_t)b * NP;
    float v[HEAD_SM_MAX];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < HEAD_SM_MAX; i++) {
        const int o = i * 64 + lane;
        v[i] = o < p.A ? lg[o] : -INFINITY;
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
    if (lane == 0) p.values[b] = tanhf(lg[p.A]);
}
