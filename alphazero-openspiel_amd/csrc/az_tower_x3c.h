// az_tower_x3c.h — az_tower_x3c_kernel: az_tower_x3b_kernel's arithmetic (fp32-grade, split-fp16 operands, no output-channel
// tile for channels 48, 49) for SMALL batches: one board per FOUR waves instead of one per wave (a workgroup = one such board up
// to 256 boards, two above: az_net.hip).
// Reference computation: ResidualBlock.forward x n_blocks of Net.forward (network.py:48-64,99-104) in eval mode.
//
// With a board per wave a launch of <= 1024 boards is one round that lasts as long as ONE board's chain of 19+ convs
// (123-150 us for a 10-block net: profiles/r3_tower_vs_boards.txt), and below 256 boards three of a CU's four SIMDs idle.  That
// is every tick of a generation's thinned-out tail and every tick of a small generation (the reference Trainer asks for 500
// games).  Here the four waves of a workgroup split ONE board by output-channel tile:
//     wave 0, 1, 2: tile mt = wave (channels 16 mt .. 16 mt + 15): 9 MFMAs per k-step (hi*hi, hi*lo, lo*hi x 3 column tiles)
//     wave 3:       tiles T and X (channels 48, 49), the scratch path of their shifted sum, their epilogue
// so a conv is 135 MFMAs deep instead of 441.  Every wave reads all B (activation) fragments of the k-steps it multiplies and
// only its own A (weight) fragments; the weight stream, its two LDS buffers and the chunk protocol are x3b's (one barrier at the
// start of a chunk's last k-step; the pieces of chunk c + 2 are issued right behind it).  The waves share the board's planes:
//   * write-after-read: a wave's epilogue stores follow the barrier of k-step 14, and every wave's LAST plane read (the
//     gather k-step's B fragments, fetched during k-step 13) has returned before it enters that barrier;
//   * read-after-write: one more barrier per conv, between the epilogue stores and the next conv's first B reads.
// Every accumulator sees the same MFMAs in the same order as in az_tower_x3b_kernel and the epilogue arithmetic is the same
// code, so a board's outputs are the same BITS whichever kernel evaluates it (tests/test_fused_net.py) - the records of a
// generation cannot depend on when its tail switches kernels.
// Two boards per workgroup (BPW = 2, eight waves): the waves of one role land on one SIMD and fill each other's waits; the boards
// use the planes and scratch of x3b's waves 0 and 1, the barriers and the weight stream are the workgroup's.  512 boards: 49 -> 39 us
// (3-block net); as a replacement for x3b at 4096 boards it loses (202 vs 173 us): the T + X wave's SIMD idles three quarters of the time.
#pragma once
#include "az_tower_x3b.h"
#include "az_head_fused.h"

// what a wave of az_tower_x3c_kernel reads for k-step ks (TX: the wave of tiles T and X, which multiplies only where they are on)
template <bool IS_FIRST, int NT, bool TX> struct X3CK {
    using K = X3BK<IS_FIRST, NT>;
    static constexpr bool mine(int ks) { return !TX || K::has_t(ks); }
    static constexpr int n_a(int ks) { return TX ? ((K::has_t(ks) ? 1 : 0) + (K::has_x(ks) ? 2 : 0)) : 2; }
    static constexpr int n_b(int ks) { return mine(ks) ? K::n_b(ks) : 0; }
};

// BPW: boards per workgroup (1: four waves, one per SIMD; 2: eight waves - the two boards' waves of the same tile share a SIMD)
template <int NT, int BPW>
__global__ __launch_bounds__(256 * BPW, BPW) void az_tower_x3c_kernel(TowerParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int WAVES = 4 * BPW, FR = X3B::FR, REC2 = X3B::REC2, CK = X3B::CK, NKS = X3B::NKS, PARTS = X3B::PARTS;
    constexpr int CHUNK_S = X3B::CHUNK_S, LO_OFF = X3B::LO_OFF, S_PLANE = X3B::S_PLANE;
    constexpr float INV_SPLIT = 1.0f / 2048.0f;
    constexpr int plane_b = X3B::PLANE_B;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, l15 = lane & 15;
    const int bl = wave >> 2, role = wave & 3, tid_b = tid & 255; // board of the workgroup, role in the board, thread in the board's four waves
    const int board0 = blockIdx.x * BPW + bl;
    const int region = X3B::OFF_ACT + bl * 2 * LO_OFF; // (x3b's map: planes of its wave bl, scratch of its wave bl)
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds;
    const int trash = X3B::OFF_EPI + 2048 + tid_b * 16; // (dump slots: two boards' threads may share one)
    const int s_wave = X3B::OFF_S + bl * X3B::S_WAVE;

    { // zero the planes and the scratch, the four waves together
        uint4 z = {0, 0, 0, 0};
        for (int i = tid_b * 16; i < 2 * LO_OFF; i += 256 * 16) *(uint4 *)(lds + region + i) = z;
        for (int i = tid_b * 16; i < X3B::S_WAVE; i += 256 * 16) *(uint4 *)(lds + s_wave + i) = z;
    }
    TowerTables<NT, true, true> T;
    T.init(p, region, plane_b, lds_base, board0, q, l15);
    int (&pos_addr)[NT] = T.pos_addr, (&grow)[NT] = T.grow, (&p6_addr)[NT] = T.p6_addr;
    int (&koff)[AZ_NET_KSTEPS] = T.koff, (&ksp)[4] = T.ksp, (&koff0)[AZ_NET_K0STEPS] = T.koff0;

    auto split4 = [&](const f32x4 &v, half4 &hi, half4 &lo) { split4_f16x3(v, hi, lo); };
    __syncthreads(); // the zeroes are down before wave 0 writes the input planes

    // ---- weight stream (az_tower_x3b.h): chunk c -> buffer c & 1; the four waves issue a chunk's pieces together
    auto issue_chunk = [&](int c, auto part_c) {
        constexpr int part = decltype(part_c)::value;
        constexpr int NPIECES = (part < 0 ? X3B::C0_B : X3B::part_bytes(part < 0 ? 0 : part)) / 1024;
        const size_t off = part < 0 ? 0 : (size_t)X3B::C0_B + (size_t)((c - 1) / PARTS) * X3B::CONV_B + X3B::part_off(part < 0 ? 0 : part);
#pragma unroll
        for (int i = 0; i < (NPIECES + WAVES - 1) / WAVES; i++) {
            int piece = i * WAVES + wave; // (branch-free: past the chunk -> its last piece again; past the stream -> zero padding)
            piece = piece < NPIECES ? piece : NPIECES - 1;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const unsigned char *)p.conv_w + off + piece * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void *)(lds + (c & 1) * CHUNK_S + piece * 1024), 16, 0, 0);
        }
    };
    auto issue_piece = [&](int c, auto part_c, int i) { // piece 4 i + wave of chunk c
        constexpr int part = decltype(part_c)::value;
        constexpr int NPIECES = X3B::part_bytes(part) / 1024;
        const size_t off = (size_t)X3B::C0_B + (size_t)((c - 1) / PARTS) * X3B::CONV_B + X3B::part_off(part);
        int piece = i * WAVES + wave;
        piece = piece < NPIECES ? piece : NPIECES - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const unsigned char *)p.conv_w + off + piece * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void *)(lds + (c & 1) * CHUNK_S + piece * 1024), 16, 0, 0);
    };
    issue_chunk(0, std::integral_constant<int, -1>{});
    issue_chunk(1, std::integral_constant<int, 0>{});
    if (wave == 0)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const unsigned char *)p.epi + lane * 16),
                                         (__attribute__((address_space(3))) void *)(lds + X3B::OFF_EPI), 16, 0, 0);

    // ROLE 0..2: output-channel tile mt = ROLE.  ROLE 3: tiles T and X (channels 48, 49).
    auto body = [&](auto role_c) {
        constexpr int ROLE = decltype(role_c)::value;
        constexpr bool TX = ROLE == 3;
        constexpr int mt = ROLE; // (tile T keeps index 3 in the parameter tables)
        f32x4 acc[NT], xres[NT]; // this wave's tile (ROLE 3: tile T); ONE accumulator, 2048 x the conv (az_tower_x3b.h)
        f32x4 accx[NT];          // ROLE 3: tile X
        { // prologue: wave 0 writes a = lrelu(bn1(x0)) -> octet 0 (hi, lo); every wave takes its tile's share of the block-1 skip conv
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
                    if (ROLE == 0 && q == 0) {
                        f32x4 a;
#pragma unroll
                        for (int c = 0; c < 4; c++) a[c] = c < p.cin ? lrelu(p.in_scale[c] * v[c] + p.in_shift[c]) : 0.f;
                        half4 hi, lo;
                        split4_planes(a, hi, lo);
                        *(half4 *)(lds + pos_addr[nt]) = hi;
                        *(half4 *)(lds + pos_addr[nt] + LO_OFF) = lo;
                    }
                }
                f32x4 x;
#pragma unroll
                for (int r = 0; r < 4; r++) x[r] = sw[r][0] * v[0] + sw[r][1] * v[1] + sw[r][2] * v[2] + sw[r][3] * v[3];
                xres[nt] = x;
                acc[nt] = (!TX || q == 0) ? *(const f32x4 *)(p.epi + 16 * mt + 4 * q) * X3_WSCALE : (f32x4){0.f, 0.f, 0.f, 0.f};
                if (TX) acc[nt][2] = acc[nt][3] = 0.f;
                accx[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        // scratch addresses (ROLE 3; az_tower_x3b.h)
        int sdst[NT][2], scen[NT];
        if constexpr (TX) {
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                const int y = 2 * nt + (l15 >> 3), x = l15 & 7;
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const int t = 2 * q + k, tap = t < 4 ? t : t + 1;
                    const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
                    const int yd = y - dy, xd = x - dx;
                    const bool ok = grow[nt] >= 0 && yd >= 0 && yd < p.H && xd >= 0 && xd < p.W;
                    sdst[nt][k] = ok ? s_wave + t * S_PLANE + (yd * 8 + xd) * 8 : trash;
                }
                scen[nt] = (q == 1 && grow[nt] >= 0) ? s_wave + 8 * S_PLANE + (nt * 16 + l15) * 8 : trash;
            }
        }
        const unsigned sread = lds_base + s_wave + l15 * 8;

        half8 ah0, al0, at0; // A fragments of a conv's k-step 0 (x3b: fetched during the last k-step of the conv before)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads(); // chunks 0, 1 and the parameters have landed; the input planes are written
        {
            const unsigned wb0 = lds_base + lane * 16;
            if constexpr (!TX) {
                READ_A(ah0, wb0, mt * FR);
                READ_A(al0, wb0, (3 + mt) * FR);
            } else READ_A(at0, wb0, AZ_NET_K0STEPS * REC2);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr (!TX) {
                launder(ah0);
                launder(al0);
            } else launder(at0);
        }

        int chunk = 0;
        auto conv_step = [&](int conv, const auto &kf, auto is_first_c) {
            constexpr bool IS_FIRST = decltype(is_first_c)::value;
            constexpr int NKSC = IS_FIRST ? AZ_NET_K0STEPS : NKS;
            using K = X3BK<IS_FIRST, NT>;
            half8 ah[2], al[2], at[2], axh[2], axl[2];
            half8 bh[2][NT], bl[2][NT];
            unsigned sph[NT][4], spl[NT][4];
            f32x4 ep_sc, ep_sh, ep_nb;
            const unsigned ep_base = lds_base + X3B::OFF_EPI + (conv & 1) * 1024 + q * 16;
            using C = X3CK<IS_FIRST, NT, TX>;
            auto read_a = [&](unsigned wb, auto buf_c, auto ks_c, auto r_c) {
                constexpr int buf = decltype(buf_c)::value, ks = decltype(ks_c)::value, r = decltype(r_c)::value;
                constexpr int part = IS_FIRST ? 0 : ks / CK, ksl = ks - part * CK;
                if constexpr (!TX) {
                    if constexpr (r == 0) READ_A(ah[buf], wb, ksl * REC2 + mt * FR);
                    else READ_A(al[buf], wb, ksl * REC2 + (3 + mt) * FR);
                } else {
                    constexpr int xbase = IS_FIRST ? AZ_NET_K0STEPS * REC2 + ks * FR : (part == 1 ? 4 * REC2 + (ks - 6) * 3 * FR : 3 * REC2);
                    if constexpr (r == 0) READ_A(at[buf], wb, xbase);
                    else if constexpr (r == 1) READ_A(axh[buf], wb, xbase + FR);
                    else READ_A(axl[buf], wb, xbase + 2 * FR);
                }
            };
            auto read_b = [&](auto buf_c, auto ks_c, auto r_c) {
                constexpr int buf = decltype(buf_c)::value, ks = decltype(ks_c)::value, r = decltype(r_c)::value;
                if constexpr (K::is_gather(ks)) {
                    constexpr bool lo = r >= 4 * NT;
                    constexpr int nt = (r % (4 * NT)) / 4, i = r % 4;
                    if constexpr (lo) READ_B32_OFF(spl[nt][i], (unsigned)ksp[i], nt * 64 + LO_OFF);
                    else READ_B32_OFF(sph[nt][i], (unsigned)ksp[i], nt * 64);
                } else {
                    constexpr int nt = r % NT;
                    if constexpr (r >= NT) READ_B_OFF(bl[buf][nt], (unsigned)kf[ks], nt * 256 + LO_OFF);
                    else READ_B_OFF(bh[buf][nt], (unsigned)kf[ks], nt * 256);
                }
            };
            f32x2 s49[NT];
            asm volatile("" ::: "memory");
            static_for<C::n_b(0)>([&](auto r_c) { read_b(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, r_c); });
            static_for<NKSC>([&](auto ks_c) {
                constexpr int ks = decltype(ks_c)::value;
                constexpr int part = IS_FIRST ? 0 : ks / CK, ksl = ks - part * CK;
                constexpr int CKL = IS_FIRST ? AZ_NET_K0STEPS : (part == PARTS - 1 ? NKS - part * CK : CK);
                constexpr bool last_of_chunk = ksl == CKL - 1, last_of_conv = ks == NKSC - 1;
                constexpr int cur = ks & 1, nxt = cur ^ 1;
                constexpr int na_next = last_of_conv ? (TX ? 0 : 2) : C::n_a(ks + 1);
                constexpr int n_next = last_of_conv ? na_next : na_next + C::n_b(ks + 1);
                constexpr bool T_ON = TX && K::has_t(ks), X_ON = TX && K::has_x(ks), GATHER = K::is_gather(ks);
                constexpr int NM = TX ? (T_ON ? 2 * NT : 0) + (X_ON ? 3 * NT : 0) : 3 * NT;
                constexpr int part2 = IS_FIRST ? 1 : (part + 2) % PARTS;
                const unsigned wb_cur = lds_base + (chunk & 1) * CHUNK_S + lane * 16, wb_oth = lds_base + ((chunk + 1) & 1) * CHUNK_S + lane * 16;
                const unsigned wb_next = last_of_chunk ? wb_oth : wb_cur;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if constexpr (last_of_chunk) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    if (!IS_FIRST && part == 0 && wave == 0)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const unsigned char *)p.epi + (size_t)conv * 1024 + lane * 16),
                                                         (__attribute__((address_space(3))) void *)(lds + X3B::OFF_EPI + (conv & 1) * 1024), 16, 0, 0);
                    if constexpr (TX) issue_chunk(chunk + 2, std::integral_constant<int, part2>{}); // buffer chunk & 1 is free (this wave is idle anyway)
                }
                // The tile waves spread their 8 pieces of chunk + 2 over three k-steps, one piece per three MFMAs - a burst of 8
                // costs a 9-MFMA k-step more than it multiplies: pieces 0..2 here (d = 0), 3..5 and 6, 7 in the first two k-steps of
                // the next chunk (d = 1, 2; the chunk counter has moved on by then, and so has the part)
                constexpr int d = last_of_chunk ? 0 : (ksl < 2 && !IS_FIRST ? ksl + 1 : -1);
                constexpr int part_t = d == 0 ? part2 : (part + 1) % PARTS; // part of the chunk the pieces belong to
                const int chunk_t = d == 0 ? chunk + 2 : chunk + 1;
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (last_of_conv) {
                    if constexpr (!IS_FIRST) {
                        lds_read_f4_off<256 + mt * 64>(ep_sc, ep_base);
                        lds_read_f4_off<512 + mt * 64>(ep_sh, ep_base);
                    }
                    lds_read_f4_off<768 + mt * 64>(ep_nb, ep_base);
                }
                // the reads of the next k-step (all of them up front when this wave has nothing to multiply here)
                auto next_read = [&](auto r_c) {
                    constexpr int r = decltype(r_c)::value;
                    if constexpr (last_of_conv) {
                        if constexpr (r == 0) READ_A(ah[nxt], wb_next, mt * FR);
                        else READ_A(al[nxt], wb_next, (3 + mt) * FR);
                    } else if constexpr (r < na_next)
                        read_a(wb_next, std::integral_constant<int, nxt>{}, std::integral_constant<int, ks + 1>{}, r_c);
                    else read_b(std::integral_constant<int, nxt>{}, std::integral_constant<int, ks + 1>{}, std::integral_constant<int, r - na_next>{});
                };
                if constexpr (NM == 0) static_for<n_next>(next_read);
                // a k-step is only 9 MFMAs deep here: the next k-step's reads go out in its first slots (three per slot; one or eight per slot measured the same), so that
                // the last of them has most of the k-step to land before the next lgkmcnt(0)
                constexpr int NMD = NM ? NM : 1;
                constexpr int RPS = (n_next + NMD - 1) / NMD > 3 ? (n_next + NMD - 1) / NMD : 3;
                static_for<NM>([&](auto j_c) {
                    constexpr int j = decltype(j_c)::value;
                    static_for<RPS>([&](auto rr_c) {
                        constexpr int r = RPS * j + decltype(rr_c)::value;
                        if constexpr (r < n_next) next_read(std::integral_constant<int, r>{});
                    });
                    if constexpr (!TX && d >= 0 && j % 3 == 2 && 3 * d + j / 3 < (30 + WAVES - 1) / WAVES)
                        issue_piece(chunk_t, std::integral_constant<int, part_t>{}, 3 * d + j / 3);
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
                    auto b_hi = [&](auto nt_c) -> half8 {
                        constexpr int nt = decltype(nt_c)::value;
                        if constexpr (GATHER) {
                            const u32x4 u = {sph[nt][0], sph[nt][1], sph[nt][2], sph[nt][3]};
                            return __builtin_bit_cast(half8, u);
                        } else return bh[cur][nt];
                    };
                    auto b_lo = [&](auto nt_c) -> half8 {
                        constexpr int nt = decltype(nt_c)::value;
                        if constexpr (GATHER) {
                            const u32x4 u = {spl[nt][0], spl[nt][1], spl[nt][2], spl[nt][3]};
                            return __builtin_bit_cast(half8, u);
                        } else return bl[cur][nt];
                    };
                    if constexpr (!TX) { // one accumulator: hi'*hi, hi'*lo0, lo*hi (the order of az_tower_x3b_kernel per accumulator and k-step)
                        constexpr int pass = j / NT, nt = j % NT;
                        constexpr auto ntc = std::integral_constant<int, nt>{};
                        const half8 a_hi = ks == 0 ? ah0 : ah[cur], a_lo = ks == 0 ? al0 : al[cur];
                        if constexpr (pass == 0) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, b_hi(ntc), acc[nt], 0, 0, 0);
                        else if constexpr (pass == 1) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, b_lo(ntc), acc[nt], 0, 0, 0);
                        else acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo, b_hi(ntc), acc[nt], 0, 0, 0);
                    } else if constexpr (T_ON && j < 2 * NT) {
                        constexpr int nt = j % NT;
                        constexpr auto ntc = std::integral_constant<int, nt>{};
                        const half8 a_t = ks == 0 ? at0 : at[cur];
                        if constexpr (j < NT) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_t, b_hi(ntc), acc[nt], 0, 0, 0);
                        else acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_t, b_lo(ntc), acc[nt], 0, 0, 0);
                    } else {
                        constexpr int jj = j - 2 * NT, nt = jj % NT;
                        constexpr auto ntc = std::integral_constant<int, nt>{};
                        if constexpr (jj < NT) accx[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(axh[cur], b_hi(ntc), ks == 6 ? zero4 : accx[nt], 0, 0, 0);
                        else if constexpr (jj < 2 * NT) accx[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(axh[cur], b_lo(ntc), accx[nt], 0, 0, 0);
                        else accx[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(axl[cur], b_hi(ntc), accx[nt], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
                // channels 48, 49: tile X and the centre-tap rows of T are final after k-step 7; this wave has nothing to multiply
                // until the gather k-step, so the shifted sum through the scratch runs here in one piece (the same arithmetic, in the
                // same order, as x3b's interleaved version)
                if constexpr (TX && !IS_FIRST && ks == 8) {
#pragma unroll
                    for (int nt = 0; nt < NT; nt++) {
                        f32x4 xv;
#pragma unroll
                        for (int i = 0; i < 4; i++) xv[i] = accx[nt][i] * INV_SPLIT;
                        f32x2 cv;
#pragma unroll
                        for (int i = 0; i < 2; i++) cv[i] = (acc[nt][i] + acc[nt][i + 2]) * INV_SPLIT;
                        lds_write64(lds_base + sdst[nt][0], (f32x2){xv[0], xv[1]});
                        lds_write64(lds_base + sdst[nt][1], (f32x2){xv[2], xv[3]});
                        lds_write64(lds_base + scen[nt], cv);
                    }
                    static_for<NT>([&](auto nt_c) {
                        constexpr int nt = decltype(nt_c)::value;
                        f32x2 pl[9];
                        static_for<9>([&](auto t_c) { lds_read64_off<decltype(t_c)::value * S_PLANE + nt * 128>(pl[decltype(t_c)::value], sread); });
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        static_for<9>([&](auto t_c) { launder(pl[decltype(t_c)::value]); });
                        f32x2 s = pl[0];
#pragma unroll
                        for (int t = 1; t < 9; t++) s = s + pl[t];
                        s49[nt] = s;
                    });
                }
                if constexpr (last_of_chunk) chunk++;
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr (!TX) { // the next conv's k-step-0 fragments outlive the epilogue: launder (az_net_common.h)
                constexpr int slot = NKSC & 1;
                launder(ah[slot]);
                launder(al[slot]);
                ah0 = ah[slot];
                al0 = al[slot];
            }
            if constexpr (!IS_FIRST) {
                launder(ep_sc);
                launder(ep_sh);
            }
            launder(ep_nb);
            // ---- epilogue of this wave's tile, in fp32; the result is split into (hi, lo) again (az_tower_x3b.h, same arithmetic)
            auto epilogue = [&](auto kind) {
                constexpr int KIND = decltype(kind)::value; // 0: conv1, 1: conv2 (not last), 2: last conv
                const int co0 = 16 * mt + 4 * q;
                const int woff = (2 * mt + (q >> 1)) * plane_b + (q & 1) * 8;
                const f32x4 sc = ep_sc, sh = ep_sh, next_bias = ep_nb;
#pragma unroll
                for (int nt = 0; nt < NT; nt++) {
                    f32x4 v;
                    if constexpr (!TX) {
                        v = acc[nt] * INV_SPLIT;
                        acc[nt] = next_bias; // (2048 x the next conv's bias: scaled on the host)
                    } else {
                        v = (f32x4){(acc[nt][0] + acc[nt][2]) * INV_SPLIT, (acc[nt][1] + acc[nt][3]) * INV_SPLIT, 0.f, 0.f};
                        if constexpr (!IS_FIRST) {
                            v[0] += s49[nt][0];
                            v[1] += s49[nt][1];
                        }
                        if (q != 0) v = (f32x4){0.f, 0.f, 0.f, 0.f};
                        acc[nt] = q == 0 ? (f32x4){next_bias[0], next_bias[1], 0.f, 0.f} : (f32x4){0.f, 0.f, 0.f, 0.f};
                    }
                    f32x4 o;
                    if (KIND == 0) {
                        o = __builtin_elementwise_max(v, v * 0.01f);
                    } else {
                        f32x4 xv = xres[nt] + v;
                        xres[nt] = xv;
                        if (KIND == 2) {
                            half4 hi, lo;
                            split4(xv, hi, lo);
                            if (grow[nt] >= 0 && co0 < p.xout_c) {
                                *(half4 *)(p.xout + (size_t)grow[nt] * p.xout_c + co0) = hi;
                                *(half4 *)(p.xout_lo + (size_t)grow[nt] * p.xout_c + co0) = lo;
                            }
                            if (!p.fc_w) continue;
                            o = xv; // fused head: the tower output also goes into the planes (fc1 below reads it from there)
                        } else {
                            f32x4 a = __builtin_elementwise_fma(sc, xv, sh);
                            o = __builtin_elementwise_max(a, a * 0.01f);
                        }
                    }
                    half4 hi, lo;
                    if (KIND == 2) split4(o, hi, lo); // (the tower OUTPUT in the planes, for the fused head: its format, lo x 2048)
                    else split4_planes(o, hi, lo);    // between convs the lo half is unscaled (az_net_common.h)
                    if constexpr (TX) {
                        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                        const bool live = q == 0 && grow[nt] >= 0;
                        *(unsigned *)(lds + (live ? p6_addr[nt] : trash)) = __builtin_bit_cast(u32x2, hi)[0];
                        *(unsigned *)(lds + (live ? p6_addr[nt] + LO_OFF : trash + 8)) = __builtin_bit_cast(u32x2, lo)[0];
                    } else {
                        const bool live = grow[nt] >= 0;
                        *(half4 *)(lds + (live ? pos_addr[nt] + woff : trash)) = hi;
                        *(half4 *)(lds + (live ? pos_addr[nt] + woff + LO_OFF : trash + 8)) = lo;
                    }
                }
            };
            if constexpr (IS_FIRST) epilogue(std::integral_constant<int, 0>{});
            else {
                if (!(conv & 1)) epilogue(std::integral_constant<int, 0>{});
                else if (conv != p.n_convs - 1) epilogue(std::integral_constant<int, 1>{});
                else epilogue(std::integral_constant<int, 2>{});
            }
            __syncthreads(); // the other waves' tiles of the new activations are in the planes before anybody reads them
        };
        conv_step(0, koff0, std::true_type{});
        for (int conv = 1; conv < p.n_convs; conv++) conv_step(conv, koff, std::false_type{});
    };
    if (role == 0) body(std::integral_constant<int, 0>{});
    else if (role == 1) body(std::integral_constant<int, 1>{});
    else if (role == 2) body(std::integral_constant<int, 2>{});
    else body(std::integral_constant<int, 3>{});

    // ---- fc1 + softmax + tanh for the workgroup's board(s), when the net has a single output tile (az_head_fused.h): the weight
    // buffers are free now (every wave's last __syncthreads drained its LDS-DMA); one board: two chains per wave; two boards: one
    // chain per wave for both (rows 0-7 / 8-15)
    if (p.fc_w) {
        if constexpr (BPW == 1) x3_fused_head<4, 4, true>(p, lds, lds, region, 0, plane_b, LO_OFF, lane, role, role == 0 ? 0 : -1, board0);
        else x3_fused_head<3, 8, false>(p, lds, lds, X3B::OFF_ACT, 2 * LO_OFF, plane_b, LO_OFF, lane, wave, role == 0 ? bl : -1, board0);
    }
}
