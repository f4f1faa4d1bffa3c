// az_tower_x3.h — az_tower_x3_kernel: the same tower at fp32-grade precision, split-fp16 operands (AZ_NET_PREC_F16X3).
#pragma once
#include "az_net_common.h"

// ------------------------------------------------------------------------------------------------
// az_tower_x3_kernel - the same tower at fp32-GRADE precision on the fp16 matrix pipe ("f16x3", precision AZ_NET_PREC_F16X3).
//
// The reference's Net.forward is fp32 (network.py:48-64).  gfx950's f32-input MFMA runs at 1/16 of the f16 rate, so instead
// every operand is carried as TWO fp16 numbers, x = hi + lo / 2048 with hi = fp16(x), lo = fp16((x - hi) * 2048) (the scale
// keeps lo out of the fp16 subnormals), and a product is three MFMAs with fp32 accumulation:
//     acc  += W_hi * A_hi                      (exact products, 22-bit)
//     acc2 += W_hi * A_lo + W_lo * A_hi        (scaled by 2048; the dropped W_lo * A_lo term is ~2^-22 relative)
//     result = acc + acc2 / 2048
// i.e. ~22 mantissa bits per product against fp32's 24, at 3/16 of the cost of the f32 MFMA path.  Measured against an fp64
// evaluation of the same net the error is of the order of torch-fp32's own (tests/test_fused_net.py).
//
// Structure: one workgroup = 4 waves (one per SIMD: 36-48 MFMAs per k-step hide the LDS latency without a second wave),
// one board per wave; the activation image has a hi and a lo set of channel-octet planes; the weight stream carries, per
// k-step, a hi record followed by a lo record (both in the f16 kernel's record format); epilogues run in fp32 and split
// their result again.  Tables, tile shapes, the 15-k-step grouping and the LDS-DMA double buffer are those of
// az_tower_kernel above.

template <int NT, int CK, bool RP1, int R3>
__global__ __launch_bounds__(256, 1) void az_tower_x3_kernel(TowerParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int WAVES = 4;
    constexpr int REC = WRec<R3>::BYTES, REC2 = 2 * REC; // one k-step: hi record, lo record
    constexpr int CHUNK_B = CK * REC2;
    constexpr int CHUNK_S = (CHUNK_B + 1023) & ~1023;
    constexpr bool L15 = R3 < 16;
    constexpr int NKS = L15 ? 15 : AZ_NET_KSTEPS;
    constexpr float INV_SPLIT = 1.0f / 2048.0f, SPLIT = 2048.0f;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, l15 = lane & 15;
    const int plane_b = p.rcells * OCT_B, region_b = N_OCT * plane_b;
    const int lo_off = RP1 ? X3_LOFF_RP1 : region_b;
    const int board0 = blockIdx.x * WAVES + wave;
    const int region = p.off_act + wave * 2 * lo_off;
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds;
    const int trash = p.off_epi + 2048 + tid * 16; // per-thread dump slot (hi at +0, lo at +8) for masked-out stores

    { // zero both plane sets (halo + padding must read as 0)
        uint4 z = {0, 0, 0, 0};
        for (int i = lane * 16; i < 2 * lo_off; i += 64 * 16) *(uint4 *)(lds + region + i) = z;
    }
    TowerTables<NT, RP1, L15> T; // per-lane address tables (az_net_common.h); one board per wave here
    T.init(p, region, plane_b, lds_base, board0, q, l15);
    int (&pos_addr)[NT] = T.pos_addr, (&grow)[NT] = T.grow, (&p6_addr)[NT] = T.p6_addr;
    int (&koff)[AZ_NET_KSTEPS] = T.koff, (&ksp)[4] = T.ksp, (&koff0)[AZ_NET_K0STEPS] = T.koff0;

    // x -> (hi, lo): hi = fp16(x), lo = fp16((x - hi) * 2048)
    auto split4 = [&](const f32x4 &v, half4 &hi, half4 &lo) {
        hi = __builtin_convertvector(v, half4);
        lo = __builtin_convertvector((v - __builtin_convertvector(hi, f32x4)) * SPLIT, half4);
    };

    f32x4 acc[4][NT], acc2[4][NT], xres[4][NT];
    { // prologue: a = lrelu(bn1(x0)) -> octet 0 (hi, lo); block-1 skip conv3(x0) in fp32 -> residual stream
        f32x4 sw[4][4];
#pragma unroll
        for (int mt = 0; mt < 4; mt++)
#pragma unroll
            for (int r = 0; r < 4; r++) sw[mt][r] = *(const f32x4 *)(p.skip_w + (16 * mt + 4 * q + r) * 4);
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (grow[nt] >= 0) {
                int gb = grow[nt] / p.HW, pos = grow[nt] - gb * p.HW;
#pragma unroll
                for (int c = 0; c < 4; c++)
                    if (c < p.cin) v[c] = p.obs[((size_t)gb * p.cin + c) * p.HW + pos];
                if (q == 0) {
                    f32x4 a;
#pragma unroll
                    for (int c = 0; c < 4; c++) a[c] = c < p.cin ? lrelu(p.in_scale[c] * v[c] + p.in_shift[c]) : 0.f;
                    half4 hi, lo;
                    split4(a, hi, lo);
                    *(half4 *)(lds + pos_addr[nt]) = hi;
                    *(half4 *)(lds + pos_addr[nt] + lo_off) = lo;
                }
            }
#pragma unroll
            for (int mt = 0; mt < 4; mt++) {
                f32x4 x;
#pragma unroll
                for (int r = 0; r < 4; r++)
                    x[r] = sw[mt][r][0] * v[0] + sw[mt][r][1] * v[1] + sw[mt][r][2] * v[2] + sw[mt][r][3] * v[3];
                xres[mt][nt] = x;
                acc[mt][nt] = *(const f32x4 *)(p.epi + 16 * mt + 4 * q);
                acc2[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    }

    constexpr int PARTS = (NKS + CK - 1) / CK;
    constexpr int C0_B = AZ_NET_K0STEPS * REC2;
    static_assert(CK % 2 == 0 && AZ_NET_K0STEPS % 2 == 0, "fragment buffer parity relies on an even chunk length");
    static_assert((PARTS & (PARTS - 1)) == 0 && NKS - (PARTS - 1) * CK >= 3, "chunk index arithmetic / the last two k-steps share a chunk");
    static_assert((CK - 1) * REC2 + REC + 4 * 1024 <= 65536, "A-fragment offsets must fit the ds offset field");
    static_assert(C0_B <= CHUNK_B && REC % 16 == 0, "conv 0 must fit a chunk buffer");
    const int n_chunks = 1 + (p.n_convs - 1) * PARTS;
    auto issue_bytes = [&](const unsigned char *src, unsigned char *dst, auto bytes_c) {
        constexpr int NPIECES = (decltype(bytes_c)::value + 1023) / 1024;
#pragma unroll
        for (int i = 0; i < (NPIECES + WAVES - 1) / WAVES; i++) {
            int piece = i * WAVES + wave;
            if (piece < NPIECES)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + piece * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void *)(dst + piece * 1024), 16, 0, 0);
        }
    };
    auto issue_chunk = [&](int c) {
        const int ci = (c - 1) / PARTS, part = (c - 1) & (PARTS - 1);
        issue_bytes((const unsigned char *)p.conv_w + C0_B + ((size_t)ci * NKS + (size_t)part * CK) * REC2, lds + (c & 1) * CHUNK_S,
                    std::integral_constant<int, CHUNK_B>{});
    };
    issue_bytes((const unsigned char *)p.conv_w, lds, std::integral_constant<int, C0_B>{});
    if (wave == 0)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const unsigned char *)p.epi + lane * 16),
                                         (__attribute__((address_space(3))) void *)(lds + p.off_epi), 16, 0, 0);

    int chunk = 0;
    auto conv_step = [&](int conv, const auto &kf, auto is_first_c) {
        constexpr bool IS_FIRST = decltype(is_first_c)::value;
        constexpr int NPARTS = IS_FIRST ? 1 : PARTS;
        constexpr int NKSC = IS_FIRST ? AZ_NET_K0STEPS : NKS;
        constexpr bool HAS_SPECIAL = L15 && !IS_FIRST;
        half8 ah[2][4], al[2][4], bh[2][NT], bl[2][NT]; // hi / lo fragments, double buffered over k-steps
        unsigned sph[NT][4], spl[NT][4];                // gather k-step: B fragments dword by dword
        f32x4 ep_sc[4], ep_sh[4], ep_nb[4];
        const unsigned ep_base = lds_base + p.off_epi + (conv & 1) * 1024 + q * 16;
        static_for<NPARTS>([&](auto part_c) {
            constexpr int part = decltype(part_c)::value;
            constexpr int CKL = part == NPARTS - 1 ? NKSC - part * CK : CK;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const unsigned wbl = lds_base + (chunk & 1) * CHUNK_S + lane * 16;
            const unsigned wbl3 = R3 == 16 ? wbl
                                           : lds_base + (chunk & 1) * CHUNK_S +
                                                 (q * WRec<R3>::ROWS + (l15 < WRec<R3>::ROWS - 1 ? l15 : WRec<R3>::ROWS - 1)) * 16;
            // read r of k-step ks (compile-time) into fragment buffer `buf`.  Read order inside a k-step:
            //   A_hi 0..3, A_lo 0..3, then B: plain k-step B_hi 0..NT-1, B_lo 0..NT-1; gather k-step 4 dwords per tile, hi then lo
            auto read_a = [&](auto buf_c, auto ksl_c, auto r_c) {
                constexpr int buf = decltype(buf_c)::value, ksl = decltype(ksl_c)::value, r = decltype(r_c)::value;
                constexpr int mt = r & 3;
                if constexpr (r < 4) READ_A(ah[buf][mt], mt < 3 ? wbl : wbl3, ksl * REC2 + mt * 1024);
                else READ_A(al[buf][mt], mt < 3 ? wbl : wbl3, ksl * REC2 + REC + mt * 1024);
            };
            auto read_b = [&](auto buf_c, auto ks_c, auto r_c) { // r in [0, 2 NT)
                constexpr int buf = decltype(buf_c)::value, ks = decltype(ks_c)::value, r = decltype(r_c)::value;
                constexpr int nt = r % NT;
                constexpr bool lo = r >= NT;
                if constexpr (RP1) {
                    if constexpr (lo) READ_B_OFF(bl[buf][nt], (unsigned)kf[ks], nt * 256 + X3_LOFF_RP1);
                    else READ_B_OFF(bh[buf][nt], (unsigned)kf[ks], nt * 256);
                } else {
                    if constexpr (lo) READ_B(bl[buf][nt], lds_base + pos_addr[nt] + lo_off + opaque(kf[ks]));
                    else READ_B(bh[buf][nt], lds_base + pos_addr[nt] + opaque(kf[ks]));
                }
            };
            auto read_sp = [&](auto r_c) { // r in [0, 8 NT): tile-major, hi then lo, 4 dwords each
                constexpr int r = decltype(r_c)::value;
                constexpr bool lo = r >= 4 * NT;
                constexpr int nt = (r % (4 * NT)) / 4, i = r % 4;
                if constexpr (RP1) {
                    if constexpr (lo) READ_B32_OFF(spl[nt][i], (unsigned)ksp[i], nt * 64 + X3_LOFF_RP1);
                    else READ_B32_OFF(sph[nt][i], (unsigned)ksp[i], nt * 64);
                } else {
                    if constexpr (lo) READ_B32_OFF(spl[nt][i], lds_base + p6_addr[nt] + lo_off + opaque(ksp[i]), 0);
                    else READ_B32_OFF(sph[nt][i], lds_base + p6_addr[nt] + opaque(ksp[i]), 0);
                }
            };
            static_for<8>([&](auto r_c) { read_a(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, r_c); });
            if constexpr (part == 0) // later chunks of a conv had their B fragments fetched before the barrier
                static_for<2 * NT>([&](auto r_c) { read_b(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, r_c); });
            // the other weight buffer is free now: fetch the next chunk (issued after the fragment reads so that their
            // latency hides behind the DMA issue)
            if (chunk + 1 < n_chunks) issue_chunk(chunk + 1);
            if (!IS_FIRST && part == 0 && wave == 0)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const unsigned char *)p.epi + (size_t)conv * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void *)(lds + p.off_epi + (conv & 1) * 1024), 16, 0, 0);
            static_for<CKL>([&](auto ksl_c) {
                constexpr int ksl = decltype(ksl_c)::value, ksg = part * CK + ksl;
                constexpr int cur = ksl & 1, nxt = cur ^ 1;
                constexpr bool more_here = ksl + 1 < CKL;
                constexpr bool more_next = !more_here && part + 1 < NPARTS;
                constexpr bool cur_gather = HAS_SPECIAL && ksg == NKSC - 1;
                constexpr bool next_gather = HAS_SPECIAL && ksg + 1 == NKSC - 1;
                constexpr int n_b_next = next_gather ? 8 * NT : 2 * NT;
                constexpr int n_next = more_here ? 8 + n_b_next : (more_next ? n_b_next : 0);
                constexpr int ks_next = (more_here || more_next) ? ksg + 1 : 0;
                constexpr bool last_of_conv = !more_here && !more_next;
                constexpr int NM = 3 * 4 * NT; // MFMAs of this k-step
                static_assert(n_next <= NM, "one read of the next k-step per MFMA slot");
                // every fragment of this k-step was issued at least (NM - n_next) MFMAs ago (or just after the chunk barrier)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (last_of_conv)
                    static_for<4>([&](auto mt_c) {
                        constexpr int mt = decltype(mt_c)::value;
                        if constexpr (!IS_FIRST) {
                            lds_read_f4_off<256 + mt * 64>(ep_sc[mt], ep_base);
                            lds_read_f4_off<512 + mt * 64>(ep_sh[mt], ep_base);
                        }
                        lds_read_f4_off<768 + mt * 64>(ep_nb[mt], ep_base);
                    });
                static_for<NM>([&](auto j_c) {
                    constexpr int j = decltype(j_c)::value;
                    constexpr int pass = j / (4 * NT), nt = (j % (4 * NT)) >> 2, mt = j & 3;
                    if constexpr (j < n_next) { // read j of the next k-step, in its read order
                        constexpr int r = more_here ? j : j + 8; // a B-only prefetch skips the A slots
                        if constexpr (r < 8) read_a(std::integral_constant<int, nxt>{}, std::integral_constant<int, ksl + 1>{}, std::integral_constant<int, r>{});
                        else if constexpr (next_gather) read_sp(std::integral_constant<int, r - 8>{});
                        else read_b(std::integral_constant<int, nxt>{}, std::integral_constant<int, ks_next>{}, std::integral_constant<int, r - 8>{});
                    }
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    half8 bhi, blo;
                    if constexpr (cur_gather) {
                        const u32x4 uh = {sph[nt][0], sph[nt][1], sph[nt][2], sph[nt][3]};
                        const u32x4 ul = {spl[nt][0], spl[nt][1], spl[nt][2], spl[nt][3]};
                        bhi = __builtin_bit_cast(half8, uh);
                        blo = __builtin_bit_cast(half8, ul);
                    } else {
                        bhi = bh[cur][nt];
                        blo = bl[cur][nt];
                    }
                    if constexpr (pass == 0) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[cur][mt], bhi, acc[mt][nt], 0, 0, 0);
                    else if constexpr (pass == 1) acc2[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[cur][mt], blo, acc2[mt][nt], 0, 0, 0);
                    else acc2[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[cur][mt], bhi, acc2[mt][nt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
            chunk++;
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        static_for<4>([&](auto mt_c) {
            constexpr int mt = decltype(mt_c)::value;
            if constexpr (!IS_FIRST) {
                keep_alive(ep_sc[mt]);
                keep_alive(ep_sh[mt]);
            }
            keep_alive(ep_nb[mt]);
        });
        // ---- epilogue, in fp32; the result is split into (hi, lo) again ------------------------------------------
        auto epilogue = [&](auto kind) {
            constexpr int KIND = decltype(kind)::value; // 0: conv1, 1: conv2 (not last), 2: last conv
#pragma unroll
            for (int mt = 0; mt < 4; mt++) {
                const int co0 = 16 * mt + 4 * q;
                const bool wr = (2 * mt + (q >> 1)) < N_OCT;
                const int woff = (2 * mt + (q >> 1)) * plane_b + (q & 1) * 8;
                const f32x4 sc = ep_sc[mt], sh = ep_sh[mt], next_bias = ep_nb[mt];
#pragma unroll
                for (int nt = 0; nt < NT; nt++) {
                    f32x4 v = acc[mt][nt] + acc2[mt][nt] * INV_SPLIT;
                    acc[mt][nt] = next_bias * INV_SPLIT; // (the ring holds 2048 x the bias, for the single-accumulator kernels)
                    acc2[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    f32x4 o;
                    if (KIND == 0) {
                        o = __builtin_elementwise_max(v, v * 0.01f);
                    } else {
                        f32x4 xv = xres[mt][nt] + v;
                        xres[mt][nt] = xv;
                        if (KIND == 2) {
                            half4 hi, lo;
                            split4(xv, hi, lo);
                            if (grow[nt] >= 0 && co0 < p.xout_c) {
                                *(half4 *)(p.xout + (size_t)grow[nt] * p.xout_c + co0) = hi;
                                *(half4 *)(p.xout_lo + (size_t)grow[nt] * p.xout_c + co0) = lo;
                            }
                            continue;
                        }
                        f32x4 a = __builtin_elementwise_fma(sc, xv, sh);
                        o = __builtin_elementwise_max(a, a * 0.01f);
                    }
                    half4 hi, lo;
                    split4(o, hi, lo);
                    if (L15 && mt == 3) { // channels 48, 49 -> the compact planes
                        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                        const bool live = q == 0 && grow[nt] >= 0;
                        *(unsigned *)(lds + (live ? p6_addr[nt] : trash)) = __builtin_bit_cast(u32x2, hi)[0];
                        *(unsigned *)(lds + (live ? p6_addr[nt] + lo_off : trash + 8)) = __builtin_bit_cast(u32x2, lo)[0];
                    } else {
                        const bool live = wr && grow[nt] >= 0;
                        *(half4 *)(lds + (live ? pos_addr[nt] + woff : trash)) = hi;
                        *(half4 *)(lds + (live ? pos_addr[nt] + woff + lo_off : trash + 8)) = lo;
                    }
                }
            }
        };
        if constexpr (IS_FIRST) epilogue(std::integral_constant<int, 0>{});
        else {
            if (!(conv & 1)) epilogue(std::integral_constant<int, 0>{});
            else if (conv != p.n_convs - 1) epilogue(std::integral_constant<int, 1>{});
            else epilogue(std::integral_constant<int, 2>{});
        }
    };
    conv_step(0, koff0, std::true_type{});
    for (int conv = 1; conv < p.n_convs; conv++) conv_step(conv, koff, std::false_type{});
}
