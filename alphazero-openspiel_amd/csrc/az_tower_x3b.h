// az_tower_x3b.h — az_tower_x3b_kernel: the fp32-grade (split-fp16, AZ_NET_PREC_F16X3) tower for row-pair boards
// (W <= 7, H <= 6: connect_four, breakthrough up to 6x6) with <= 50 filters, WITHOUT an output-channel tile for channels 48, 49.
// Reference computation: ResidualBlock.forward x n_blocks of Net.forward (network.py:48-64,99-104) in eval mode.
//
// az_tower_x3_kernel (az_tower_x3.h) spends a whole 16-row output-channel tile - a quarter of its MFMAs - on the two
// channels 48, 49: every row of an MFMA sees the same B operand, so with (tap, input channel) on K and output channels on
// M the 14 other rows of that tile are dead.  Here, for those two output channels only, the roles of tap and M are swapped:
//
//     out[c, p] = sum_tap sum_ci W[c, tap, ci] A[ci, p + d(tap)]           c = 48, 49
//               = sum_tap D[(c, tap), p + d(tap)],     D[(c, tap), p'] = sum_ci W[c, tap, ci] A[ci, p']
//
// D has (c, tap) on its ROWS and the UNSHIFTED activations as its B operand - exactly the B fragments the k-steps of
// the centre tap (k-steps 6 and 7 of the 15-k-step grouping: groups (tap 4, octet 0..5)) already hold in registers.  So:
//   * tile X (16 rows = the 8 off-centre taps x 2 channels) is multiplied in k-steps 6, 7 only; its columns are then
//     SHIFTED by d(tap) and summed in the epilogue through a small wave-private fp32 scratch in LDS (S: 9 planes of
//     48 positions x 2 channels; a lane stores its two taps' values at the DESTINATION position, the lanes that own
//     channels 48, 49 read the nine planes at their own position; entries whose source lies off the board are never
//     written and stay 0 from the prologue = the conv's zero padding);
//   * tile T (the old tile 3) keeps rows 0..3 for the gather k-step (input channels 48, 49 of all nine taps - their
//     B fragment IS tap-shifted - rows: hi 48, hi 49, lo 48, lo 49) and gets rows 4..7 for the centre tap of D (no
//     shift; hi, hi, lo, lo), so it is multiplied in k-steps 6, 7 and 14 only, twice (x B_hi, x B_lo) instead of three
//     times: with the hi and lo weights on different ROWS one MFMA yields hi*hi and lo*hi together.
// MFMAs per column tile and conv: 15 x 9 (tiles 0..2, three per product) + 2 x (2 T + 3 X) + 2 T = 147 against 180: -18 %.
//
// Since round 4 a tile has ONE accumulator (az_net_common.h: split_pair_planes): the device copy of the weights carries 2048 in both
// halves and the planes keep the lo half unscaled, so the three MFMAs of a product add into the same registers; an accumulator
// holds 2048 x the conv, the epilogue scales it back (exactly).
// Everything else is az_tower_x3_kernel's: one workgroup = 4 waves (one per SIMD), one board per wave, hi and lo
// activation planes, fp32 residual stream in registers, fp32 epilogues that split their result into (hi, lo) again.
// The compact plane of channels 48, 49 takes its real 4 bytes per cell (the scratch S lives in what that frees).
#pragma once
#include "az_net_common.h"

// what k-step ks (index in its conv) multiplies besides tiles 0..2, and how many fragment reads it needs
template <bool IS_FIRST, int NT> struct X3BK {
    static constexpr bool has_t(int ks) { return IS_FIRST || ks == 6 || ks == 7 || ks == X3B::NKS - 1; }
    static constexpr bool has_x(int ks) { return !IS_FIRST && (ks == 6 || ks == 7); }
    static constexpr bool is_gather(int ks) { return !IS_FIRST && ks == X3B::NKS - 1; }
    static constexpr int n_a(int ks) { return 6 + (has_t(ks) ? 1 : 0) + (has_x(ks) ? 2 : 0); } // ah 0..2, al 0..2, T, X hi, X lo
    static constexpr int n_b(int ks) { return is_gather(ks) ? 8 * NT : 2 * NT; }
};

template <int NT>
__global__ __launch_bounds__(256, 1) void az_tower_x3b_kernel(TowerParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int WAVES = 4, FR = X3B::FR, REC2 = X3B::REC2, CK = X3B::CK, NKS = X3B::NKS, PARTS = X3B::PARTS;
    constexpr int CHUNK_S = X3B::CHUNK_S, LO_OFF = X3B::LO_OFF, S_PLANE = X3B::S_PLANE;
    constexpr float INV_SPLIT = 1.0f / 2048.0f;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, l15 = lane & 15;
    constexpr int plane_b = X3B::PLANE_B;
    const int board0 = blockIdx.x * WAVES + wave;
    const int region = X3B::OFF_ACT + wave * 2 * LO_OFF;
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds;
    const int trash = X3B::OFF_EPI + 2048 + tid * 16; // per-thread dump slot (hi at +0, lo at +8) for masked-out stores
    const int s_wave = X3B::OFF_S + wave * X3B::S_WAVE;

    { // zero both plane sets (halo + padding must read as 0) and the scratch (entries without an on-board source stay 0)
        uint4 z = {0, 0, 0, 0};
        for (int i = lane * 16; i < 2 * LO_OFF; i += 64 * 16) *(uint4 *)(lds + region + i) = z;
        for (int i = lane * 16; i < X3B::S_WAVE; i += 64 * 16) *(uint4 *)(lds + s_wave + i) = z;
    }
    TowerTables<NT, true, true> T; // per-lane address tables (az_net_common.h); one board per wave
    T.init(p, region, plane_b, lds_base, board0, q, l15);
    int (&pos_addr)[NT] = T.pos_addr, (&grow)[NT] = T.grow, (&p6_addr)[NT] = T.p6_addr;
    int (&koff)[AZ_NET_KSTEPS] = T.koff, (&ksp)[4] = T.ksp, (&koff0)[AZ_NET_K0STEPS] = T.koff0;

    // scratch addresses.  Lane (q, l15) of tile X holds rows 4q..4q+3 = (plane 2q, c0), (2q, c1), (2q+1, c0), (2q+1, c1) at
    // position (y, x) = (2 nt + (l15 >> 3), l15 & 7); plane t belongs to tap tap_of_plane(t) with d = (dy, dx): the value is
    // a term of out[c, (y - dy, x - dx)].  Off-board destinations and padding lanes store to the trash slot.
    int sdst[NT][2], scen[NT];
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
    const unsigned sread = lds_base + s_wave + l15 * 8; // plane t, tile nt: + t * S_PLANE + nt * 128

    // x -> (hi, lo): hi = fp16(x), lo = fp16((x - hi) * 2048)
    auto split4 = [&](const f32x4 &v, half4 &hi, half4 &lo) { split4_f16x3(v, hi, lo); }; // (five instructions per pair: az_net_common.h)

    f32x4 acc[4][NT], xres[4][NT]; // [3]: tile T.  The accumulators hold 2048 x the conv (X3_WSCALE)
    f32x4 accx[NT];                // tile X
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
                    split4_planes(a, hi, lo);
                    *(half4 *)(lds + pos_addr[nt]) = hi;
                    *(half4 *)(lds + pos_addr[nt] + LO_OFF) = lo;
                }
            }
#pragma unroll
            for (int mt = 0; mt < 4; mt++) {
                f32x4 x;
#pragma unroll
                for (int r = 0; r < 4; r++)
                    x[r] = sw[mt][r][0] * v[0] + sw[mt][r][1] * v[1] + sw[mt][r][2] * v[2] + sw[mt][r][3] * v[3];
                xres[mt][nt] = x;
                // tile T: only channels 48, 49 (lanes q == 0, rows 0, 1) carry a bias; its other rows are lo / centre-tap rows
                acc[mt][nt] = (mt < 3 || q == 0) ? *(const f32x4 *)(p.epi + 16 * mt + 4 * q) * X3_WSCALE : (f32x4){0.f, 0.f, 0.f, 0.f};
                if (mt == 3) acc[mt][nt][2] = acc[mt][nt][3] = 0.f;
            }
            accx[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }

    // ---- weight stream: chunk c -> buffer c & 1 by LDS-DMA (global_load_lds, one KiB per wave-instruction).  Chunk 0 = conv 0,
    // chunk c >= 1 = part (c - 1) % 4 of conv 1 + (c - 1) / 4.  A chunk's ONE barrier sits at the start of its LAST k-step: by
    // then every wave holds all of the chunk's fragments in registers (its buffer is free: chunk c + 2 is fetched into it, a piece
    // every few MFMAs), and the vmcnt(0) in front of the barrier makes chunk c + 1 - issued a whole chunk earlier - visible, so
    // that k-step's prefetch of the NEXT k-step's fragments already reads the other buffer.  The MFMA stream therefore runs
    // through chunk boundaries without a bubble; only the conv boundary (epilogue, then the B fragments of k-step 0) breaks it.
    auto issue_piece = [&](int c, auto part_c, int i) { // piece 4 i + wave of chunk c (= part `part` of its conv; part -1: conv 0)
        constexpr int part = decltype(part_c)::value;
        constexpr int NPIECES = (part < 0 ? X3B::C0_B : X3B::part_bytes(part < 0 ? 0 : part)) / 1024;
        // branch-free: a wave whose piece index runs past the chunk re-fetches the chunk's last piece (same bytes, same place),
        // and a chunk index past the stream fetches its zero padding (az_net.hip)
        int piece = i * WAVES + wave;
        piece = piece < NPIECES ? piece : NPIECES - 1;
        const size_t off = part < 0 ? 0 : (size_t)X3B::C0_B + (size_t)((c - 1) / PARTS) * X3B::CONV_B + X3B::part_off(part < 0 ? 0 : part);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const unsigned char *)p.conv_w + off + piece * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void *)(lds + (c & 1) * CHUNK_S + piece * 1024), 16, 0, 0);
    };
    constexpr int NPW = (CHUNK_S / 1024 + WAVES - 1) / WAVES; // pieces per wave of the largest chunk
#pragma unroll
    for (int i = 0; i < NPW; i++) issue_piece(0, std::integral_constant<int, -1>{}, i);
#pragma unroll
    for (int i = 0; i < NPW; i++) issue_piece(1, std::integral_constant<int, 0>{}, i);
    if (wave == 0)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const unsigned char *)p.epi + lane * 16),
                                         (__attribute__((address_space(3))) void *)(lds + X3B::OFF_EPI), 16, 0, 0);
    half8 ah0[3], al0[3], at0; // A fragments of a conv's k-step 0: fetched during the LAST k-step of the conv before (the k-step
                               // count is odd, so the two-deep fragment ring cannot carry them across the conv boundary)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    {
        const unsigned wb0 = lds_base + lane * 16;
        static_for<3>([&](auto r_c) {
            constexpr int r = decltype(r_c)::value;
            READ_A(ah0[r], wb0, r * FR);
            READ_A(al0[r], wb0, (3 + r) * FR);
        });
        READ_A(at0, wb0, AZ_NET_K0STEPS * REC2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        static_for<3>([&](auto r_c) {
            launder(ah0[decltype(r_c)::value]);
            launder(al0[decltype(r_c)::value]);
        });
        launder(at0);
    }

    int chunk = 0; // the chunk of the current k-step
    auto conv_step = [&](int conv, const auto &kf, auto is_first_c) {
        constexpr bool IS_FIRST = decltype(is_first_c)::value;
        constexpr int NKSC = IS_FIRST ? AZ_NET_K0STEPS : NKS;
        using K = X3BK<IS_FIRST, NT>;
        half8 ah[2][3], al[2][3], at[2], axh[2], axl[2]; // A fragments (weights): tiles 0..2 hi / lo, T, X hi / lo; k-step ks >= 1 uses ring slot ks & 1
        half8 bh[2][NT], bl[2][NT];                      // B fragments (activations), hi / lo
        unsigned sph[NT][4], spl[NT][4];                 // gather k-step: B fragments dword by dword
        f32x4 ep_sc[4], ep_sh[4], ep_nb[4];
        const unsigned ep_base = lds_base + X3B::OFF_EPI + (conv & 1) * 1024 + q * 16;
        // fragment read r of k-step ks (its chunk's buffer at wb) into ring slot buf.  Order: ah 0..2, al 0..2, T, X hi, X lo
        auto read_a = [&](unsigned wb, auto buf_c, auto ks_c, auto r_c) {
            constexpr int buf = decltype(buf_c)::value, ks = decltype(ks_c)::value, r = decltype(r_c)::value;
            constexpr int part = IS_FIRST ? 0 : ks / CK, ksl = ks - part * CK;
            if constexpr (r < 3) READ_A(ah[buf][r], wb, ksl * REC2 + r * FR);
            else if constexpr (r < 6) READ_A(al[buf][r - 3], wb, ksl * REC2 + r * FR);
            else {
                // the extra fragments sit behind the chunk's records
                constexpr int xbase = IS_FIRST ? AZ_NET_K0STEPS * REC2 + ks * FR : (part == 1 ? 4 * REC2 + (ks - 6) * 3 * FR : 3 * REC2);
                if constexpr (r == 6) READ_A(at[buf], wb, xbase);
                else if constexpr (r == 7) READ_A(axh[buf], wb, xbase + FR);
                else READ_A(axl[buf], wb, xbase + 2 * FR);
            }
        };
        auto read_b = [&](auto buf_c, auto ks_c, auto r_c) { // r in [0, n_b): plain: hi 0..NT-1, lo 0..NT-1; gather: tile-major dwords
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
        // channels 48, 49 (az_tower_x3b.h header): tile X and the centre-tap rows of tile T are final after k-step 7, so their
        // shifted sum through the scratch runs INSIDE the k-loop, a few instructions per MFMA slot of k-steps 8..12:
        //   k-step 8: combine (hi, lo) accumulators, store every value at its destination position;
        //   k-steps 9, 10, 11 (slots 9..17): read the nine tap planes of column tile 0, 1, 2 (waited for by the next k-step's lgkmcnt(0));
        //   k-steps 10, 11, 12 (slots 0..8): sum them.  The epilogue adds the gather k-step's part (tile T rows 0..3) and finishes.
        f32x2 s49[NT]; // (lanes q == 0) sum over the nine tap planes at this lane's position: channels 48, 49
        f32x2 pl[9];
        f32x4 xv;
        f32x2 cv;
        auto s_path = [&](auto ks_c, auto j_c) {
            constexpr int ks = decltype(ks_c)::value, j = decltype(j_c)::value;
            if constexpr (IS_FIRST) return;
            if constexpr (ks == 8 && j < 9 * NT) { // 9 slots per column tile: 4 combine, 2 centre, 3 stores
                constexpr int nt = j / 9, i = j % 9;
                if constexpr (i < 4) xv[i] = accx[nt][i] * INV_SPLIT;
                else if constexpr (i < 6) // (lanes q == 1 hold rows 4..7 of tile T: hi c0, hi c1, lo c0, lo c1 of the centre tap)
                    cv[i - 4] = (acc[3][nt][i - 4] + acc[3][nt][i - 2]) * INV_SPLIT;
                else if constexpr (i == 6) lds_write64(lds_base + sdst[nt][0], (f32x2){xv[0], xv[1]});
                else if constexpr (i == 7) lds_write64(lds_base + sdst[nt][1], (f32x2){xv[2], xv[3]});
                else lds_write64(lds_base + scen[nt], cv);
            }
            if constexpr (ks >= 10 && ks <= 12 && j < 9) { // the planes of tile nt, read in the k-step before, have landed (k-step start wait)
                constexpr int nt = ks - 10, t = j;
                if constexpr (t == 0) {
                    static_for<9>([&](auto t_c) { launder(pl[decltype(t_c)::value]); });
                    s49[nt] = pl[0];
                } else s49[nt] = s49[nt] + pl[t];
            }
            if constexpr (ks >= 9 && ks <= 11 && j >= 9 && j < 18) { // (LDS operations of one wave execute in order: these reads see the stores)
                constexpr int nt = ks - 9, t = j - 9;               // issued after the sums of the tile before, which still hold pl
                lds_read64_off<t * S_PLANE + nt * 128>(pl[t], sread);
            }
        };
        asm volatile("" ::: "memory"); // (the epilogue's LDS stores stay above these untracked reads)
        static_for<K::n_b(0)>([&](auto r_c) { read_b(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, r_c); });
        static_for<NKSC>([&](auto ks_c) {
            constexpr int ks = decltype(ks_c)::value;
            constexpr int part = IS_FIRST ? 0 : ks / CK, ksl = ks - part * CK;
            constexpr int CKL = IS_FIRST ? AZ_NET_K0STEPS : (part == PARTS - 1 ? NKS - part * CK : CK);
            constexpr bool last_of_chunk = ksl == CKL - 1, last_of_conv = ks == NKSC - 1;
            constexpr int cur = ks & 1, nxt = cur ^ 1;
            constexpr int na_next = last_of_conv ? 6 : K::n_a(ks + 1); // across the conv boundary: ah0 / al0 of the next conv
            constexpr int n_next = last_of_conv ? 6 : na_next + K::n_b(ks + 1);
            constexpr bool T_ON = K::has_t(ks), X_ON = K::has_x(ks), GATHER = K::is_gather(ks);
            constexpr int NM = 9 * NT + (T_ON ? 2 * NT : 0) + (X_ON ? 3 * NT : 0);
            constexpr int RPS = (n_next + NM - 1) / NM > 1 ? (n_next + NM - 1) / NM : 1; // reads per MFMA slot (2 or 3 per slot measured 1-3 % slower)
            constexpr int part2 = IS_FIRST ? 1 : (part + 2) % PARTS;                      // the part chunk + 2 is
            static_assert(2 + 3 * (NPW - 1) < NM, "a DMA piece every third MFMA slot");
            const unsigned wb_cur = lds_base + (chunk & 1) * CHUNK_S + lane * 16, wb_oth = lds_base + ((chunk + 1) & 1) * CHUNK_S + lane * 16;
            const unsigned wb_next = last_of_chunk ? wb_oth : wb_cur; // where the next k-step's fragments live
            // every fragment of this k-step was issued early in the previous one
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr (last_of_chunk) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (!IS_FIRST && part == 0 && wave == 0) // this conv's epilogue parameters ride the same DMA path into a 2-slot ring
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const unsigned char *)p.epi + (size_t)conv * 1024 + lane * 16),
                                                     (__attribute__((address_space(3))) void *)(lds + X3B::OFF_EPI + (conv & 1) * 1024), 16, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!IS_FIRST && ks == 8) // (tile X is final: keep it in its AGPRs until the scratch path reads it, below)
                static_for<NT>([&](auto nt_c) {
                    pin_acc(accx[decltype(nt_c)::value]);
                });
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
                static_for<RPS>([&](auto rr_c) { // reads of the next k-step, in its read order
                    constexpr int r = RPS * j + decltype(rr_c)::value;
                    if constexpr (r < n_next) {
                        if constexpr (last_of_conv) { // k-step 0 of the next conv (records start at the head of its first chunk)
                            if constexpr (r < 3) READ_A(ah[nxt][r], wb_next, r * FR);
                            else READ_A(al[nxt][r - 3], wb_next, r * FR);
                        } else if constexpr (r < na_next)
                            read_a(wb_next, std::integral_constant<int, nxt>{}, std::integral_constant<int, ks + 1>{}, std::integral_constant<int, r>{});
                        else read_b(std::integral_constant<int, nxt>{}, std::integral_constant<int, ks + 1>{}, std::integral_constant<int, r - na_next>{});
                    }
                });
                if constexpr (last_of_chunk && j >= 2 && (j - 2) % 3 == 0 && (j - 2) / 3 < NPW) // buffer chunk & 1 is free: fetch chunk + 2
                    issue_piece(chunk + 2, std::integral_constant<int, part2>{}, (j - 2) / 3);
                s_path(ks_c, j_c);
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
                // ONE accumulator per tile: the weights carry 2048 in both halves and the activations' lo half is unscaled, so hi'*hi,
                // hi'*lo0 and lo*hi are all 2048 x their share of the product (az_net_common.h: split_pair_planes).  Order per accumulator
                // and k-step - hi'*hi, hi'*lo0, lo*hi - is the same in az_tower_x3c_kernel and az_tower_x3d_kernel: same bits.
                if constexpr (j < 9 * NT) { // tiles 0..2, pass-major over the (tile, column tile) pairs
                    constexpr int pass = j / (3 * NT), nt = (j % (3 * NT)) / 3, mt = j % 3;
                    constexpr auto ntc = std::integral_constant<int, nt>{};
                    const half8 a_hi = ks == 0 ? ah0[mt] : ah[cur][mt], a_lo = ks == 0 ? al0[mt] : al[cur][mt];
                    if constexpr (pass == 0) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, b_hi(ntc), acc[mt][nt], 0, 0, 0);
                    else if constexpr (pass == 1) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, b_lo(ntc), acc[mt][nt], 0, 0, 0);
                    else acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo, b_hi(ntc), acc[mt][nt], 0, 0, 0);
                } else if constexpr (T_ON && j < 11 * NT) { // tile T (hi rows and lo rows in one fragment): x B_hi, then x B_lo0
                    constexpr int jj = j - 9 * NT, nt = jj % NT;
                    constexpr auto ntc = std::integral_constant<int, nt>{};
                    const half8 a_t = ks == 0 ? at0 : at[cur];
                    if constexpr (jj < NT) acc[3][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_t, b_hi(ntc), acc[3][nt], 0, 0, 0);
                    else acc[3][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_t, b_lo(ntc), acc[3][nt], 0, 0, 0);
                } else { // tile X: hi'*hi (from a literal 0 at k-step 6), hi'*lo0, lo*hi
                    constexpr int jj = j - 11 * NT, nt = jj % NT;
                    constexpr auto ntc = std::integral_constant<int, nt>{};
                    if constexpr (jj < NT) accx[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(axh[cur], b_hi(ntc), ks == 6 ? zero4 : accx[nt], 0, 0, 0);
                    else if constexpr (jj < 2 * NT) accx[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(axh[cur], b_lo(ntc), accx[nt], 0, 0, 0);
                    else accx[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(axl[cur], b_hi(ntc), accx[nt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (last_of_chunk) chunk++;
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // The next conv's k-step-0 fragments outlive the epilogue.  An untracked asynchronous read looks "defined" to the
        // compiler the moment it is issued, so a long-lived value may be copied (e.g. parked in an AGPR) BEFORE its data has
        // arrived: pass the ring registers through an asm placed after the wait, and only then hand them to ah0 / al0.
        static_for<3>([&](auto r_c) {
            constexpr int r = decltype(r_c)::value, slot = NKSC & 1; // the ring slot the last k-step prefetched into
            launder(ah[slot][r]);
            launder(al[slot][r]);
            ah0[r] = ah[slot][r];
            al0[r] = al[slot][r];
        });
        static_for<4>([&](auto mt_c) { // (same for the prefetched epilogue parameters)
            constexpr int mt = decltype(mt_c)::value;
            if constexpr (!IS_FIRST) {
                launder(ep_sc[mt]);
                launder(ep_sh[mt]);
            }
            launder(ep_nb[mt]);
        });
        // The accumulators live in AGPRs (the kernel holds ~470 registers), the epilogue's arithmetic needs them in VGPRs, and left to
        // itself the register allocator makes that copy right behind the MFMA that produces the final value: every MFMA of the last
        // k-step was followed by the wait for its own result (s_nop 7 + four v_accvgpr_read_b32, 33 times per conv).  Pinned in their
        // AGPRs here, the values are copied where the epilogue uses them - long after the MFMAs have drained.
#pragma unroll
        for (int mt = 0; mt < 4; mt++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                pin_acc(acc[mt][nt]);
            }
        // ---- epilogue, in fp32; the result is split into (hi, lo) again ------------------------------------------
        auto epilogue = [&](auto kind) {
            constexpr int KIND = decltype(kind)::value; // 0: conv1, 1: conv2 (not last), 2: last conv
#pragma unroll
            for (int mt = 0; mt < 4; mt++) {
                const int co0 = 16 * mt + 4 * q;
                const int woff = (2 * mt + (q >> 1)) * plane_b + (q & 1) * 8;
                const f32x4 sc = ep_sc[mt], sh = ep_sh[mt], next_bias = ep_nb[mt];
#pragma unroll
                for (int nt = 0; nt < NT; nt++) {
                    f32x4 v;
                    if (mt < 3) {
                        v = acc[mt][nt] * INV_SPLIT;
                        acc[mt][nt] = next_bias; // (2048 x the next conv's bias: scaled on the host)
                    } else { // tile T, lanes q == 0: rows hi 48, hi 49, lo 48, lo 49 of the gather k-step (+ bias), plus the tap planes
                        v = (f32x4){(acc[3][nt][0] + acc[3][nt][2]) * INV_SPLIT, (acc[3][nt][1] + acc[3][nt][3]) * INV_SPLIT, 0.f, 0.f};
                        if constexpr (!IS_FIRST) {
                            v[0] += s49[nt][0];
                            v[1] += s49[nt][1];
                        }
                        if (q != 0) v = (f32x4){0.f, 0.f, 0.f, 0.f}; // (rows 4..15: centre-tap rows / unused)
                        acc[3][nt] = q == 0 ? (f32x4){next_bias[0], next_bias[1], 0.f, 0.f} : (f32x4){0.f, 0.f, 0.f, 0.f};
                    }
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
                    split4_planes(o, hi, lo); // (between convs the lo half is unscaled; the tower OUTPUT above keeps lo x 2048: the head's format)
                    if (mt == 3) { // channels 48, 49 -> the compact planes
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
