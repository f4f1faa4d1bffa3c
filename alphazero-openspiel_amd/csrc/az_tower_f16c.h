// az_tower_f16c.h — az_tower_f16c_kernel: az_tower_kernel's arithmetic (fp16 MFMA operands, AZ_NET_PREC_F16) for SMALL batches:
// one board per WORKGROUP of four waves instead of one per wave (what az_tower_x3c.h is for the fp32-grade tower).
// Reference computation: ResidualBlock.forward x n_blocks of Net.forward (network.py:48-64,99-104) in eval mode.
//
// With a board per wave a launch of <= 512 boards lasts as long as ONE board's chain of convs (63-66 us for a 10-block net,
// profiles/r3_tower_vs_boards.txt) on a quarter of the chip.  Here wave mt of the workgroup takes output-channel tile mt
// (channels 16 mt .. 16 mt + 15; tile 3 = channels 48, 49) for the board's three column tiles: 3 MFMAs per k-step and wave, the B
// (activation) fragments read by all four waves, the A (weight) fragments only by their owner.  Row-pair boards with <= 50
// filters only (connect_four, breakthrough up to 6x6: NT = 3, the 15-k-step grouping, the compact record of tile 3).
// The accumulator of (mt, nt) sees the same MFMAs in the same order as in az_tower_kernel, initialised with the same bias, and
// the epilogue is the same code: the same BITS whichever kernel evaluates a board (tests/test_fused_net.py).
// The waves share the board's planes: a barrier before the epilogue (every wave's B reads of this conv are done), and the next
// chunk barrier stands between the epilogue's stores and the next conv's reads; the weight stream and its record layout are
// az_tower_kernel's.  All fragment reads of a weight chunk are issued at its top and every k-step's MFMAs wait with a count.
// Measured (256 boards, 10 blocks, tower + head): 62.8 -> 47 us; 512 boards (16-KiB chunks, two workgroups per CU) 66 -> 59 us.
// The slope is 1.7 us per conv whatever was tried on the schedule - reads left to the compiler 47.3, hoisted 43.5-47, a four-deep
// weight ring 45.6, one body for the four waves instead of four specialised copies 46.9, a per-workgroup rotation of the weight
// pieces 47.6 (box-to-box noise is of that size).  Not a shared resource either: with half the workgroups (128 boards) the x3c
// launch is 8 % shorter, and two boards per workgroup at 256 boards are slower (77 vs 65 us): it is one workgroup's own chain of
// barriers, LDS round trips and dependent MFMAs per conv.
#pragma once
#include "az_tower_f16.h"

// RING: weight-chunk buffers in LDS: chunk c + RING - 1 is requested while chunk c is multiplied; the waits are counted (vmcnt) and
// the barriers bare, as in az_head_gemm_kernel.
template <int NT, int CK, int RING>
__global__ __launch_bounds__(256, 1) void az_tower_f16c_kernel(TowerParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int R3 = 2, WAVES = 4, NKS = 15;
    constexpr int REC = WRec<R3>::BYTES, ROWS = WRec<R3>::ROWS;
    constexpr int CHUNK_B = CK * REC, CHUNK_S = CK * 4 * 64 * 16;
    constexpr int PARTS = (NKS + CK - 1) / CK, C0_B = AZ_NET_K0STEPS * REC;
    static_assert(PARTS >= 2, "a conv's epilogue parameters are requested one conv = at least two chunk steps ahead (see the order of issue below)");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, l15 = lane & 15;
    const int plane_b = p.rcells * OCT_B, region_b = N_OCT * plane_b;
    const int board0 = blockIdx.x, region = p.off_act;
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds;
    const int trash = p.off_epi + 2048 + tid * 8;

    { // zero the planes, the four waves together (halo + padding must read as 0)
        uint4 z = {0, 0, 0, 0};
        for (int i = tid * 16; i < region_b; i += 256 * 16) *(uint4 *)(lds + region + i) = z;
    }
    TowerTables<NT, true, true> T;
    T.init(p, region, plane_b, lds_base, board0, q, l15);
    int (&pos_addr)[NT] = T.pos_addr, (&grow)[NT] = T.grow, (&p6_addr)[NT] = T.p6_addr;
    int (&koff)[AZ_NET_KSTEPS] = T.koff, (&ksp)[4] = T.ksp, (&koff0)[AZ_NET_K0STEPS] = T.koff0;
    __syncthreads(); // the zeroes are down before wave 0 writes the input planes

    const int n_chunks = 1 + (p.n_convs - 1) * PARTS;
    // Every wave issues the SAME number of vector-memory operations per chunk (PER; a piece index past the chunk re-fetches its
    // last piece) and per parameter block (one: all four waves fetch it, the same bytes), so that one count fits all waves.
    constexpr int PER = ((CHUNK_B + 1023) / 1024 + WAVES - 1) / WAVES;
    static_assert((RING - 2) * PER < 64, "the counted waits fit vmcnt");
    auto issue_bytes = [&](const unsigned char *src, unsigned char *dst, auto bytes_c) {
        constexpr int NPIECES = (decltype(bytes_c)::value + 1023) / 1024;
#pragma unroll
        for (int i = 0; i < PER; i++) {
            int piece = i * WAVES + wave;
            piece = piece < NPIECES ? piece : NPIECES - 1;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + piece * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void *)(dst + piece * 1024), 16, 0, 0);
        }
    };
    auto issue_chunk = [&](int c) { // chunk c -> buffer c % RING
        if (c == 0) issue_bytes((const unsigned char *)p.conv_w, lds, std::integral_constant<int, C0_B>{});
        else {
            const int ci = (c - 1) / PARTS, part = (c - 1) % PARTS;
            issue_bytes((const unsigned char *)p.conv_w + C0_B + ((size_t)ci * NKS + (size_t)part * CK) * REC, lds + (c % RING) * CHUNK_S,
                        std::integral_constant<int, CHUNK_B>{});
        }
    };
    auto issue_epi = [&](int conv) { // [4][64] floats of conv's epilogue -> slot conv & 1 of the ring
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const unsigned char *)p.epi + (size_t)conv * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void *)(lds + p.off_epi + (conv & 1) * 1024), 16, 0, 0);
    };
    // The parameters of conv v are requested at the first chunk of conv v - 1 (conv 0's here): the counted waits leave the operations
    // of the last two steps in flight, and conv v's epilogue must find its parameters landed.  (Its ring slot v & 1 was last read in
    // conv v - 2's epilogue, which every wave has left when the first barrier of conv v - 1 opens.)
    issue_chunk(0);
    issue_epi(0);
#pragma unroll
    for (int c = 1; c < RING - 1; c++)
        if (c < n_chunks) issue_chunk(c);

    // ONE body for the four waves, the tile index a run-time (wave-uniform) value: four specialised copies of the unrolled convs
    // would be four times the instruction-cache footprint for four waves that then share nothing.
    {
        const int mt = wave;
        f32x4 acc[NT], xres[NT];
        { // prologue: wave 0 writes a = lrelu(bn1(x0)) -> octet 0; every wave takes its tile's share of the block-1 skip conv
            f32x4 sw[4];
#pragma unroll
            for (int r = 0; r < 4; r++) sw[r] = *(const f32x4 *)(p.skip_w + (16 * mt + 4 * q + r) * 4);
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (grow[nt] >= 0) {
                    int gb = grow[nt] / p.HW, pos = grow[nt] - gb * p.HW;
#pragma unroll
                    for (int c = 0; c < 4; c++)
                        if (c < p.cin) v[c] = p.obs[((size_t)gb * p.cin + c) * p.HW + pos];
                    if (mt == 0 && q == 0) {
                        half4 a4;
#pragma unroll
                        for (int c = 0; c < 4; c++) a4[c] = c < p.cin ? (_Float16)lrelu(p.in_scale[c] * v[c] + p.in_shift[c]) : (_Float16)0;
                        *(half4 *)(lds + pos_addr[nt]) = a4;
                    }
                }
                f32x4 x;
#pragma unroll
                for (int r = 0; r < 4; r++) x[r] = sw[r][0] * v[0] + sw[r][1] * v[1] + sw[r][2] * v[2] + sw[r][3] * v[3];
                xres[nt] = x;
                acc[nt] = *(const f32x4 *)(p.epi + 16 * mt + 4 * q); // bias of conv 0
            }
        }
        int chunk = 0;
        auto conv_step = [&](int conv, const auto &kf, auto is_first_c) {
            constexpr bool IS_FIRST = decltype(is_first_c)::value;
            constexpr int NPARTS = IS_FIRST ? 1 : PARTS;
            constexpr int NKSC = IS_FIRST ? AZ_NET_K0STEPS : NKS;
            static_for<NPARTS>([&](auto part_c) {
                constexpr int part = decltype(part_c)::value;
                constexpr int CKL = part == NPARTS - 1 ? NKSC - part * CK : CK;
                { // this chunk has landed once at most the (up to RING - 2) younger chunks' operations are outstanding
                    const int younger = n_chunks - 1 - chunk < RING - 2 ? n_chunks - 1 - chunk : RING - 2;
                    static_for<RING - 1>([&](auto y_c) { // (a literal operand per possible count)
                        constexpr int y = decltype(y_c)::value;
                        if (younger == y) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(y * PER) : "memory");
                    });
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // (my stores of the last epilogue / of the input planes)
                __builtin_amdgcn_s_barrier(); // everybody's pieces of this chunk are in LDS, the buffer of chunk - 1 is free, the planes are written
                asm volatile("" ::: "memory");
                if (part == 0 && conv + 1 < p.n_convs) issue_epi(conv + 1); // (a conv AHEAD: see the order of issue above)
                if (chunk + RING - 1 < n_chunks) issue_chunk(chunk + RING - 1);
                // A fragment of this wave's tile: tiles 0..2 a KiB each; tile 3 the compact record (lane (q, l15) reads stored row
                // min(l15, ROWS - 1) of its k-group, the last stored row is zero).  ALL fragment reads of the chunk go out first
                // (asm, in program order: 1 A + NT B per k-step, 1 + 4 NT for the gather k-step), then every k-step's three MFMAs wait
                // with a count for exactly the reads behind them - left to the compiler each k-step waited out two LDS round trips.
                const unsigned wba = lds_base + (chunk % RING) * CHUNK_S +
                                     (mt < 3 ? mt * 1024 + lane * 16 : 3 * 1024 + (q * ROWS + (l15 < ROWS - 1 ? l15 : ROWS - 1)) * 16);
                half8 af[CKL], bf[CKL][NT];
                unsigned bsp[NT][4];
                constexpr bool HAS_GATHER = !IS_FIRST && part == NPARTS - 1; // the conv's last k-step is in this chunk
                static_for<CKL>([&](auto ksl_c) {
                    constexpr int ksl = decltype(ksl_c)::value, ks = part * CK + ksl;
                    READ_A(af[ksl], wba, ksl * REC);
                    if constexpr (HAS_GATHER && ksl == CKL - 1) {
                        static_for<NT>([&](auto nt_c) {
                            constexpr int nt = decltype(nt_c)::value;
                            static_for<4>([&](auto i_c) { READ_B32_OFF(bsp[nt][decltype(i_c)::value], (unsigned)ksp[decltype(i_c)::value], nt * 64); });
                        });
                    } else
                        static_for<NT>([&](auto nt_c) { READ_B_OFF(bf[ksl][decltype(nt_c)::value], (unsigned)kf[ks], decltype(nt_c)::value * 256); });
                });
                constexpr int TOTAL = CKL * (1 + NT) + (HAS_GATHER ? 3 * NT : 0);
                static_for<CKL>([&](auto ksl_c) {
                    constexpr int ksl = decltype(ksl_c)::value;
                    constexpr bool gather = HAS_GATHER && ksl == CKL - 1;
                    constexpr int done = (ksl + 1) * (1 + NT) + (gather ? 3 * NT : 0); // reads up to and including this k-step's
                    wait_lgkm(TOTAL - done);
                    __builtin_amdgcn_sched_barrier(0);
                    static_for<NT>([&](auto nt_c) {
                        constexpr int nt = decltype(nt_c)::value;
                        if constexpr (gather) {
                            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                            const u32x4 u = {bsp[nt][0], bsp[nt][1], bsp[nt][2], bsp[nt][3]};
                            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[ksl], __builtin_bit_cast(half8, u), acc[nt], 0, 0, 0);
                        } else acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[ksl], bf[ksl][nt], acc[nt], 0, 0, 0);
                    });
                    __builtin_amdgcn_sched_barrier(0);
                });
                chunk++;
            });
            // this conv's epilogue parameters (landed behind the second chunk barrier; conv 0: behind its only one)
            const float *ep = (const float *)(lds + p.off_epi + (conv & 1) * 1024 + q * 16);
            const f32x4 sc = *(const f32x4 *)(ep + 64 + mt * 16), sh = *(const f32x4 *)(ep + 128 + mt * 16), next_bias = *(const f32x4 *)(ep + 192 + mt * 16);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier(); // every wave's reads of the old activations are done (a bare barrier: the weight chunks stay in flight)
            asm volatile("" ::: "memory");
            auto epilogue = [&](auto kind) { // az_tower_kernel's, for this wave's tile
                constexpr int KIND = decltype(kind)::value; // 0: conv1, 1: conv2 (not last), 2: last conv
                const int co0 = 16 * mt + 4 * q;
                const bool wr = (2 * mt + (q >> 1)) < N_OCT;
                const int woff = (2 * mt + (q >> 1)) * plane_b + (q & 1) * 8;
#pragma unroll
                for (int nt = 0; nt < NT; nt++) {
                    f32x4 v = acc[nt];
                    acc[nt] = next_bias;
                    half4 o;
                    if (KIND == 0) {
                        o = lrelu_h4(__builtin_convertvector(v, half4));
                    } else {
                        f32x4 xv = xres[nt] + v;
                        xres[nt] = xv;
                        if (KIND == 2) {
                            o = __builtin_convertvector(xv, half4);
                            if (grow[nt] >= 0 && co0 < p.xout_c) *(half4 *)(p.xout + (size_t)grow[nt] * p.xout_c + co0) = o;
                            continue;
                        }
                        o = lrelu_h4(__builtin_convertvector(__builtin_elementwise_fma(sc, xv, sh), half4));
                    }
                    if (mt == 3) { // channels 48, 49 (lanes q = 0) -> the compact plane; 50..63 do not exist
                        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                        const int wa = (q == 0 && grow[nt] >= 0) ? p6_addr[nt] : trash;
                        *(unsigned *)(lds + wa) = __builtin_bit_cast(u32x2, o)[0];
                    } else {
                        const int wa = (wr && grow[nt] >= 0) ? pos_addr[nt] + woff : trash;
                        *(half4 *)(lds + wa) = o;
                    }
                }
            };
            if constexpr (IS_FIRST) epilogue(std::integral_constant<int, 0>{});
            else {
                if (!(conv & 1)) epilogue(std::integral_constant<int, 0>{});
                else if (conv != p.n_convs - 1) epilogue(std::integral_constant<int, 1>{});
                else epilogue(std::integral_constant<int, 2>{});
            }
            // (the next chunk barrier - or the end of the kernel - stands between these stores and the next conv's reads)
        };
        conv_step(0, koff0, std::true_type{});
        for (int conv = 1; conv < p.n_convs; conv++) conv_step(conv, koff, std::false_type{});
    }
}
