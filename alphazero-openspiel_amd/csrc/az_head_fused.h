// az_head_fused.h — fc1 + softmax + tanh (network.py:61-64) INSIDE a launch of the fp32-grade tower, for nets with a single
// output tile (A + 1 <= 16: connect_four).  Used by az_tower_x3c_kernel (small batches); written for any split of the 16 MFMA rows
// between the boards of a workgroup (az_tower_x3b_kernel with four boards was tried: same bits, slower - az_net.hip).
//
// The arithmetic is az_head_kernel<true>'s, MFMA for MFMA: that kernel splits the K = H*W*64 reduction over eight waves (k-step
// ks goes to wave ks & 7, each wave one chain of MFMAs in increasing ks) and adds the eight partial tiles in wave order.  Here the
// workgroup's waves run those eight chains (8 / NWAVES each) for ALL the workgroup's boards at once: the 16 rows of the A operand
// are split between the boards (16 >> SHIFT boards of 1 << SHIFT rows; an MFMA row depends on that row's data only) and one row
// per board is kept - the same bits as the separate kernel (tests/test_fused_net.py).  The 168 KiB of fc weights pass the CU's
// vector-memory path once per workgroup.  A fragments come from the activation planes the last epilogue wrote (octets 0-5
// sixteen bytes per cell, channels 48, 49 from the compact plane, octet 7 zero).
#pragma once
#include "az_net_common.h"

// act0 / board_stride: LDS offset of the first board's planes / bytes between two boards' planes; `scratch`: 1 KiB per board of LDS
// nobody else touches any more ([8 chains][16 outputs] + 16 logits).  my_local / my_global: the board (in the workgroup / in the
// batch) whose softmax this wave runs, my_local < 0: none.  DEEP: both of a wave's chains' weight fragments in flight at once.
template <int SHIFT, int NWAVES, bool DEEP>
__device__ __forceinline__ void x3_fused_head(const TowerParams &p, unsigned char *lds, unsigned char *scratch, int act0, int board_stride, int plane_b, int lo_off,
                                              int lane, int wave, int my_local, int my_global) {
    static_assert(SHIFT >= 2 && SHIFT <= 4 && (NWAVES == 4 || NWAVES == 8), "rows per board 4..16; the eight chains split evenly over the waves");
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr int HMAX = 12; // k-steps per chain: ceil(2 * H * W / 8), boards of <= 48 cells
    const int q = lane >> 4, l15 = lane & 15;
    // A fragment of k-step ks = (cell ks >> 1, channel half ks & 1): a chain's k-steps are 8 apart, so the half - and with it this
    // lane's octet o = 4 (ks & 1) + q and which of the three cases it reads - is fixed along the chain, and the cell moves 4 columns
    // (one row wrap at most: W >= 4) per step.  Everything but the octet and the board is wave-uniform: kept in scalar registers.
    const int region_h = act0 + (l15 >> SHIFT) * board_stride;
    auto a_frag = [&](int cell, int o, half8 &a, half8 &al) {
        const unsigned char *s16 = lds + region_h + (o <= 5 ? o * plane_b + cell * 16 : 0);
        const unsigned char *s4 = lds + region_h + (o == 6 ? 6 * plane_b + cell * 4 : 0);
        const u32x4 r16 = *(const u32x4 *)s16, r16l = *(const u32x4 *)(s16 + lo_off);
        const unsigned r4 = *(const unsigned *)s4, r4l = *(const unsigned *)(s4 + lo_off);
        const u32x4 z = {0u, 0u, 0u, 0u};
        a = __builtin_bit_cast(half8, o <= 5 ? r16 : o == 6 ? (u32x4){r4, 0u, 0u, 0u} : z);
        al = __builtin_bit_cast(half8, o <= 5 ? r16l : o == 6 ? (u32x4){r4l, 0u, 0u, 0u} : z);
    };
    const int last = p.fc_ksteps - 1;
    const float bias_v = p.fc_b[l15]; // (in flight behind the weight fragments; used after the barrier below)
    // a chain's weight fragments all at once (one round trip to L2), then its MFMAs with the A fragments read one k-step ahead
    auto load_chain = [&](int w, half8 (&wv)[HMAX], half8 (&wl)[HMAX]) {
#pragma unroll
        for (int i = 0; i < HMAX; i++) {
            const int ks = w + 8 * i < last ? w + 8 * i : last; // (past the end: a valid fragment, never multiplied)
            wv[i] = *(const half8 *)(p.fc_w + ((size_t)ks * 64 + lane) * 8);
            wl[i] = *(const half8 *)(p.fc_w_lo + ((size_t)ks * 64 + lane) * 8);
        }
    };
    auto run_chain = [&](int w_v, const half8 (&wv)[HMAX], const half8 (&wl)[HMAX]) {
        const int w = __builtin_amdgcn_readfirstlane(w_v);
        f32x4 ha = {0.f, 0.f, 0.f, 0.f}, ha2 = {0.f, 0.f, 0.f, 0.f};
        const int o = 4 * (w & 1) + q;
        int x = w >> 1, y = 0; // cell of k-step w (w < 8, W >= 4: row 0)
        half8 a_nx, al_nx;
        a_frag((y + 1) * p.rs + x + 1, o, a_nx, al_nx);
#pragma unroll
        for (int i = 0; i < HMAX; i++) {
            const half8 a = a_nx, al = al_nx;
            if (i + 1 < HMAX) {
                x += 4;
                if (x >= p.W) x -= p.W, y++;
                const int yc = y < p.H ? y : p.H - 1; // (past the end: any cell of the board, never multiplied)
                a_frag((yc + 1) * p.rs + x + 1, o, a_nx, al_nx);
            }
            if (w + 8 * i <= last) {
                ha = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, wv[i], ha, 0, 0, 0);
                ha2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, wl[i], ha2, 0, 0, 0);
                ha2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, wv[i], ha2, 0, 0, 0);
            }
        }
        ha = ha + ha2 * (1.0f / 2048.0f);
        // D row 4 q + r, column l15 = output: the first row of board b is row b << SHIFT = element 0 of the lanes q = b << (SHIFT - 2)
        if ((q & ((1 << (SHIFT - 2)) - 1)) == 0) ((float *)(scratch + (q >> (SHIFT - 2)) * 1024))[w * 16 + l15] = ha[0];
    };
    {
        half8 wv0[HMAX], wl0[HMAX];
        load_chain(wave, wv0, wl0);
        if constexpr (NWAVES == 4) {
            if constexpr (DEEP) { // the second chain's fragments in flight under the first chain's MFMAs
                half8 wv1[HMAX], wl1[HMAX];
                load_chain(wave + 4, wv1, wl1);
                run_chain(wave, wv0, wl0);
                run_chain(wave + 4, wv1, wl1);
            } else {
                run_chain(wave, wv0, wl0);
                load_chain(wave + 4, wv0, wl0);
                run_chain(wave + 4, wv0, wl0);
            }
        } else run_chain(wave, wv0, wl0);
    }
    __syncthreads();
    if (my_local >= 0) {
        float *part = (float *)(scratch + my_local * 1024);
        const int sub = l15; // (lanes 16..63 repeat lanes 0..15 and store nothing)
        float v = part[sub];
#pragma unroll
        for (int w = 1; w < 8; w++) v += part[w * 16 + sub];
        float *lg = part + 128;
        if (lane < 16) lg[sub] = v + bias_v;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // (same wave: the stores are ordered before the loads below)
        float mx = -INFINITY;
        for (int o = sub; o < p.A; o += 16) mx = fmaxf(mx, lg[o]);
#pragma unroll
        for (int off = 8; off; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 16));
        float sum = 0.f;
        for (int o = sub; o < p.A; o += 16) sum += expf(lg[o] - mx);
#pragma unroll
        for (int off = 8; off; off >>= 1) sum += __shfl_xor(sum, off, 16);
        if (lane < 16 && my_global < p.n_boards) {
            float *out = p.priors + (size_t)my_global * p.A;
            for (int o = sub; o < p.A; o += 16) out[o] = expf(lg[o] - mx) / sum;
            if (sub == 0) p.values[my_global] = tanhf(lg[p.A]);
        }
    }
}
