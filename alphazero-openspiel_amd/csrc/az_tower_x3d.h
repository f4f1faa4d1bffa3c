// az_tower_x3d.h — az_tower_x3d_kernel: the fp32-grade (split-fp16, AZ_NET_PREC_F16X3) tower on PACKED column tiles.
// Reference computation: ResidualBlock.forward x n_blocks of Net.forward (network.py:48-64,99-104) in eval mode.
//
// az_tower_x3b_kernel gives every board its own column tiles: a connect_four board is 42 positions in 3 x 16 columns (12.5 % of
// every MFMA, every B fragment and every epilogue lane is padding), a 6x6 breakthrough board 36 positions in 48 columns (25 %), and
// 8x8 boards ran az_tower_x3_kernel, which still spends a whole output-channel tile on channels 48, 49 (768 MFMAs per board and
// conv).  Here a workgroup's NB boards SHARE their tiles: NB * H * W positions fill a whole number of tiles - connect_four: 8 boards,
// 336 = 21 tiles, 390 MFMAs per board and conv against 441; 6x6: 8 boards, 288 = 18 tiles, 339 against 441; 8x8: 4 boards,
// 256 = 16 tiles, 588 against 768 - on x3b's scheme for channels 48, 49 (tiles T and X) and its single accumulator per tile.
//
//   * WHICH positions make a tile is free - a column of an MFMA only has to be the same position in every k-step - so the host
//     picks them for the LDS banks (az_net.hip: x3d_layout): board b's cells sit at b R + (y + 1) rs + x + 1 in a plane (halo rows
//     and halo column(s) are zeros, so every tap shift of every position reads the conv's zero padding from memory, no masks), and
//     R, rs are such that the NB H W cells fall into the 16 residues mod 16 equally often; tile k takes the k-th position of
//     every residue, lane l15 the one of residue l15.  A ds_read_b128 of a B fragment then touches 16 different bank groups for
//     every tap (a tap moves all 16 cells by the same amount), as with x3b's row pairs.
//   * a workgroup is EIGHT waves, two per SIMD, every one within 256 registers - no accumulator in an AGPR.  (With four waves of
//     4-6 tiles the allocator copies every final value to a vector register right behind its MFMA - s_nop 7 + four
//     v_accvgpr_read_b32 after each MFMA of a conv's last k-step, measured 30 % of that kernel; az_tower_x3b_kernel had the same
//     disease: pin_acc, az_net_common.h.)  A wave owns two or three tiles whole (three output-channel tiles + T and X); a tile left
//     over is split by output-channel tile over four waves - mt 0, 1, 2 and T + X, each a role compiled on its own (selecting the A
//     fragments by a run-time branch cost 45 spilled registers).  connect_four: waves 0-3 three tiles, waves 4-7 two and a share
//     of tile 20; 6x6: every wave two and a share of tile 16 or 17; 8x8: two tiles a wave.  What made the three-tile role fit
//     was the single accumulator (a third of the accumulator registers: az_net_common.h, split_pair_planes).
//   * the waves share the planes, so a conv has two more rendezvous than x3b's: the barrier of its last k-step also separates the
//     last plane reads from the epilogue stores, and one barrier follows the epilogue.  The shifted sum of tile X goes through
//     ONE scratch for the workgroup (a term's destination column may be another wave's), behind a chunk barrier.
//   * weights: x3b's records in chunks of three k-steps (two 18-KiB LDS buffers).
//   * B fragments live in ONE register set per tile: a tile's MFMAs of a k-step run together (nine, + T and X), and its fragments of
//     the next k-step are fetched right behind them into the same registers (waits are counted: lgkmcnt(N), N = the reads issued
//     since); A fragments of tiles 0..2 keep x3b's two-slot ring, those of T and X have one set; the per-k-step address offsets
//     are compile-time constants selected per lane where they are used, not a table of registers.
// Every accumulator sees the same MFMAs in the same order as in az_tower_x3b_kernel / az_tower_x3c_kernel and the epilogue is the
// same arithmetic: on row-pair boards a board's outputs are the same BITS in all three (tests/test_fused_net.py).  The epilogue's plane
// stores need no mask here: every lane of a tile is a real position, and a board past the batch computes on zero planes.
#pragma once
#include "az_net_common.h"

template <bool IS_FIRST> struct X3DK {
    static constexpr bool has_t(int ks) { return IS_FIRST || ks == 6 || ks == 7 || ks == X3D::NKS - 1; }
    static constexpr bool has_x(int ks) { return !IS_FIRST && (ks == 6 || ks == 7); }
    static constexpr bool is_gather(int ks) { return !IS_FIRST && ks == X3D::NKS - 1; }
    static constexpr int n_tx(int ks) { return (has_t(ks) ? 1 : 0) + (has_x(ks) ? 2 : 0); } // fragments of tiles T, X hi, X lo
    // T and X have ONE register set (a ring for three k-steps of a conv would be twelve registers): where the k-step before uses
    // it too, the fetch waits for that k-step's last MFMA, and the k-step opens with a full wait (k-step 7; conv 0's k-steps 1..3)
    static constexpr bool late_tx(int ks) { return ks >= 1 && n_tx(ks) > 0 && n_tx(ks - 1) > 0; }
    static constexpr int n_a(int ks) { return 6 + (late_tx(ks) ? 0 : n_tx(ks)); } // fetched a k-step ahead: ah 0..2, al 0..2 (+ T, X hi, X lo)
    static constexpr int n_b(int ks) { return is_gather(ks) ? 8 : 2; }            // reads per tile
};

// The read schedule of one wave.  Tiles with MFMAs in k-step ks: the extra tile counts for a wave that holds an output-channel tile of it (EXM) always, for the wave that holds its T and X (EXT)
// where tile T is on.  Sequence of a k-step m (fetching for n = m + 1): A(n) one per MFMA slot of tile 0, B(n, j) behind tile j + 1's
// first MFMA, or behind the k-step's last MFMA if tile j + 1 has none.  wait_n(n, j): reads issued AFTER the B fragments of tile j for
// k-step n, up to the wait in front of that tile's MFMAs - a LOWER bound (further LDS operations in between only make
// s_waitcnt lgkmcnt(wait_n) stricter; a larger number would let the wait pass before the fragments have landed).
template <bool IS_FIRST, int NTW, bool EXM, bool EXT> struct X3DS {
    using K = X3DK<IS_FIRST>;
    static constexpr int NKSC = IS_FIRST ? AZ_NET_K0STEPS : X3D::NKS;
    static constexpr int ntb(int ks) { return NTW + ((EXM || (EXT && K::has_t(ks))) ? 1 : 0); }
    static constexpr int wait_n(int n, int j) {
        int cnt = (ntb(n) - 1 - j) * K::n_b(n); // the rest of the k-step before
        if (j >= 1 && n + 1 < NKSC) {
            cnt += K::n_a(n + 1); // this k-step, tile 0
            const int upto = j - 1 < ntb(n + 1) ? j - 1 : ntb(n + 1);
            cnt += upto * K::n_b(n + 1); // B(n + 1, 0 .. j - 2)
        }
        return cnt < 15 ? cnt : 15;
    }
    // where k-step ks's fragment r (ah 0..2, al 0..2, T, X hi, X lo) lives in its chunk's buffer
    static constexpr int a_off(int ks, int r) {
        if (IS_FIRST) return r < 6 ? (ks & 1) * X3D::REC2 + r * X3D::FR : 2 * X3D::REC2 + (ks & 1) * X3D::FR;
        const int part = X3D::part_of(ks), ksl = ks - X3D::part_ks0(part);
        if (r < 6) return ksl * X3D::REC2 + r * X3D::FR;
        return (part == 2 ? 2 * X3D::REC2 + (ks - 6) * 3 * X3D::FR : X3D::REC2) + (r - 6) * X3D::FR;
    }
};

// The variants: which tiles wave w owns, and its share of a split tile (-1: none).
template <int V> struct X3DV;
template <> struct X3DV<0> { // breakthrough 6x6-sized: 8 boards of 36 positions in 18 tiles
    static constexpr int NTILES = 18, PC = 480, RS = 8, R = 58;
    __device__ static constexpr int first(int w) { return 2 * w; }
    __device__ static constexpr int split_tile(int w) { return 16 + (w >> 2); }
    __device__ static constexpr int split_unit(int w) { return w & 3; }
};
template <> struct X3DV<1> { // 8x8: 4 boards of 64 positions in 16 tiles
    static constexpr int NTILES = 16, PC = 416, RS = 9, R = 84;
    __device__ static constexpr int first(int w) { return 2 * w; }
    __device__ static constexpr int split_tile(int) { return -1; }
    __device__ static constexpr int split_unit(int) { return 0; }
};
template <> struct X3DV<2> { // connect_four-sized: 8 boards of 42 positions in 21 tiles - waves 0-3 three tiles, waves 4-7 two and a quarter of tile 20
    static constexpr int NTILES = 21, PC = 480, RS = 8, R = 57;
    __device__ static constexpr int first(int w) { return w < 4 ? 3 * w : 12 + 2 * (w - 4); }
    __device__ static constexpr int split_tile(int w) { return w < 4 ? -1 : 20; }
};
template <int V> struct X3DG {
    static constexpr int PC = X3DV<V>::PC, NCOL = 16 * X3DV<V>::NTILES;
    static constexpr int PLANE_B = PC * OCT_B, LO_OFF = 6 * PLANE_B + PC * 4; // hi: 6 octet planes + the compact plane of channels 48, 49
    static constexpr int S_PLANE = NCOL * 8, OFF_ACT = X3D::OFF_S + 9 * S_PLANE, LDS = OFF_ACT + 2 * LO_OFF;
    static_assert(LDS <= 160 * 1024 && LO_OFF < 65536 && 9 * S_PLANE < 65536 && PC % 16 == 0, "x3d LDS budget / ds offset fields / plane stride a multiple of 256 B");
};

template <int V>
__global__ __launch_bounds__(512, 2) void az_tower_x3d_kernel(TowerParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    using G = X3DG<V>;
    using VV = X3DV<V>;
    constexpr int WAVES = 8, NKS = X3D::NKS, PARTS = X3D::PARTS;
    constexpr int CHUNK_S = X3D::CHUNK_S, LO_OFF = G::LO_OFF, S_PLANE = G::S_PLANE, plane_b = G::PLANE_B;
    constexpr float INV_SPLIT = 1.0f / 2048.0f;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, l15 = lane & 15;
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds;
    const int trash = X3D::OFF_EPI + 2048 + (tid & 255) * 16; // dump slot (hi at +0, lo at +8) for masked-out stores; a SIMD's two waves share them (write-only)

    { // zero the scratch (entries without an on-board source stay 0 = the conv's padding) and the planes (halo cells must read as 0)
        uint4 z = {0, 0, 0, 0};
        for (int i = X3D::OFF_S + tid * 16; i < G::LDS; i += 512 * 16) *(uint4 *)(lds + i) = z;
    }
    // Byte offset (tap shift + octet plane) of the lane's k-group in a k-step: a compile-time constant per (k-step, q) - selected
    // per lane where it is used (one move and three selects), NOT kept in a table: fourteen registers is what decides whether two
    // waves fit a SIMD.  `qq` is q behind an opaque asm, so that the selects stay inside the conv loop.
    auto koff_of = [](int ks, int qv, bool first) constexpr { // groups 4 ks + q < 54 = (tap, octet) = divmod(., 6); 54, 55: zero weights, any finite data
        const int g = 4 * ks + qv;
        if (first) return g < 9 ? ((g / 3 - 1) * VV::RS + (g % 3 - 1)) * OCT_B : 0; // conv 0: group g < 9 = tap g of octet 0
        const int tap = g / 6, c8 = g % 6;
        return g >= 54 ? 0 : ((tap / 3 - 1) * VV::RS + (tap % 3 - 1)) * OCT_B + c8 * plane_b;
    };
    auto ksp_of = [](int i, int qv) constexpr { // the gather k-step: element pair i of group q is tap 4 q + i (taps past the ninth carry zero weights)
        int tap = 4 * qv + i;
        tap = tap > 8 ? 8 : tap;
        return ((tap / 3 - 1) * VV::RS + (tap % 3 - 1)) * 4;
    };
    // q = 2 b1 + b0 picks one of four compile-time constants - by arithmetic: as nested ?: on the (opaque) lane-variant q the compiler
    // built exec-mask branches, four per k-step, every one of them the end of a scheduling region: 3-4 % of the kernel and 13 spilled
    // registers of the 6x6 variant.  (Bit masks instead of the multiplies: fewer registers, the same time.)
    auto sel4 = [](int qq, int c0, int c1, int c2, int c3) {
        const int b0 = qq & 1, b1 = qq >> 1;
        return c0 + b0 * (c1 - c0) + b1 * (c2 - c0) + (b0 & b1) * (c3 - c2 - c1 + c0);
    };

    auto split4 = [&](const f32x4 &v, half4 &hi, half4 &lo) { split4_f16x3(v, hi, lo); }; // x -> (hi, lo): az_net_common.h
    __syncthreads(); // the zeroes are down before the input planes are written

    // ---- weight stream: chunk -> buffer chunk & 1 by LDS-DMA (global_load_lds, one KiB per wave-instruction)
    auto issue_piece = [&](int buf, size_t off, auto npieces_c, int i) { // piece 8 i + wave of the chunk at stream offset `off`
        constexpr int NPIECES = decltype(npieces_c)::value;
        int piece = i * WAVES + wave; // (branch-free: past the chunk -> its last piece again)
        piece = piece < NPIECES ? piece : NPIECES - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const unsigned char *)p.conv_w + off + piece * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void *)(lds + buf * CHUNK_S + piece * 1024), 16, 0, 0);
    };
    constexpr int NPW = (CHUNK_S / 1024 + WAVES - 1) / WAVES; // pieces per wave of the largest chunk (eight waves: three)
    constexpr auto c0_pieces = std::integral_constant<int, X3D::C0_PART_B / 1024>{};
#pragma unroll
    for (int i = 0; i < NPW; i++) issue_piece(0, 0, c0_pieces, i);
#pragma unroll
    for (int i = 0; i < NPW; i++) issue_piece(1, X3D::C0_PART_B, c0_pieces, i);
    if (wave == 0)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const unsigned char *)p.epi + lane * 16),
                                         (__attribute__((address_space(3))) void *)(lds + X3D::OFF_EPI), 16, 0, 0);

    // NTW: tiles the wave owns whole; EX: its share of a split tile - 0 none, 1 an output-channel tile (mt = emt), 2 tiles T and X
    auto body = [&](auto ntw_c, auto ex_c, auto emt_c) {
        constexpr int NTW = decltype(ntw_c)::value, EX = decltype(ex_c)::value;
        constexpr bool EXM = EX == 1, EXT = EX == 2;
        constexpr int NTA = NTW + (EX ? 1 : 0);  // tiles whose B fragments the wave reads
        constexpr int NTT = NTW + (EXT ? 1 : 0); // tiles whose T and X this wave holds
        constexpr int emt = decltype(emt_c)::value; // (EXM) which output-channel tile of the split tile: a compile-time constant - selecting
                                                    // ah[cur][emt] by a run-time branch per MFMA cost the register allocator 45 spilled registers
        // ---- per-lane tables: the wave's tiles (nt < NTW: its own; nt = NTW: the split one) ----------------------------------
        int tile[NTA];
        unsigned baseL[NTA]; // LDS address of the lane's cell in octet plane 0 (hi)
        int grow[NTA];       // (prologue only) global board * HW + position of the lane's column, -1 = a board past the batch
        unsigned livem = 0;  // bit nt: the column of tile nt belongs to a board of the batch
#pragma unroll
        for (int nt = 0; nt < NTA; nt++) {
            tile[nt] = nt < NTW ? VV::first(wave) + nt : VV::split_tile(wave);
            const int e = p.xd_pos[tile[nt] * 16 + l15], b = e >> 8, pos = e & 255;
            const int y = pos / p.W, x = pos - y * p.W;
            const int cell = b * VV::R + (y + 1) * VV::RS + x + 1;
            baseL[nt] = lds_base + G::OFF_ACT + cell * OCT_B;
            const int gb = blockIdx.x * p.xd_nb + b;
            grow[nt] = gb < p.n_boards ? gb * p.HW + pos : -1;
            livem |= (gb < p.n_boards ? 1u : 0u) << nt;
        }
        // the lane's cell in the compact plane of channels 48, 49 (4 bytes per cell), from its octet-plane address (16 bytes per cell);
        // opaque: derived where it is used, not carried in a register per tile
        auto p6_of = [&](int nt) { return ((opaque((int)baseL[nt]) - (int)lds_base - G::OFF_ACT) >> 2) + (int)lds_base + G::OFF_ACT + 6 * plane_b; };
        // ONE accumulator per tile, holding 2048 x the conv (az_tower_x3b.h; az_net_common.h: split_pair_planes)
        f32x4 acc[3][NTW], xres[3][NTW]; // output-channel tiles 0..2 of the wave's own column tiles
        f32x4 accT[NTT], xresT[NTT];     // tile T
        f32x4 accx[NTT];                 // tile X
        f32x4 acce, xrese;               // EXM: output-channel tile emt of the split column tile
        { // prologue: a = lrelu(bn1(x0)) -> octet 0 (hi, lo); block-1 skip conv3(x0) in fp32 -> residual stream
            f32x4 sw[4][4];
#pragma unroll
            for (int mt = 0; mt < 4; mt++)
#pragma unroll
                for (int r = 0; r < 4; r++) sw[mt][r] = *(const f32x4 *)(p.skip_w + (16 * mt + 4 * q + r) * 4);
            f32x4 swe[4];
#pragma unroll
            for (int r = 0; r < 4; r++) swe[r] = *(const f32x4 *)(p.skip_w + (16 * (EXM ? emt : 3) + 4 * q + r) * 4);
            auto skip = [&](const f32x4 (&w)[4], const f32x4 &v) {
                f32x4 x;
#pragma unroll
                for (int r = 0; r < 4; r++) x[r] = w[r][0] * v[0] + w[r][1] * v[1] + w[r][2] * v[2] + w[r][3] * v[3];
                return x;
            };
            auto bias_t = [&]() { // tile T: only channels 48, 49 (lanes q == 0, rows 0, 1) carry a bias; its other rows are lo / centre-tap rows
                f32x4 b = q == 0 ? *(const f32x4 *)(p.epi + 48) * X3_WSCALE : (f32x4){0.f, 0.f, 0.f, 0.f};
                b[2] = b[3] = 0.f;
                return b;
            };
#pragma unroll
            for (int nt = 0; nt < NTA; nt++) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (grow[nt] >= 0) {
                    int gb = grow[nt] / p.HW, pos = grow[nt] - gb * p.HW;
#pragma unroll
                    for (int c = 0; c < 4; c++)
                        if (c < p.cin) v[c] = p.obs[((size_t)gb * p.cin + c) * p.HW + pos];
                    if (q == 0 && (nt < NTW || EXT)) { // (the split tile's input planes: the wave that holds its T and X)
                        f32x4 a;
#pragma unroll
                        for (int c = 0; c < 4; c++) a[c] = c < p.cin ? lrelu(p.in_scale[c] * v[c] + p.in_shift[c]) : 0.f;
                        half4 hi, lo;
                        split4_planes(a, hi, lo);
                        *(half4 *)(lds + (baseL[nt] - lds_base)) = hi;
                        *(half4 *)(lds + (baseL[nt] - lds_base) + LO_OFF) = lo;
                    }
                }
                if (nt < NTW) {
#pragma unroll
                    for (int mt = 0; mt < 3; mt++) {
                        xres[mt][nt] = skip(sw[mt], v);
                        acc[mt][nt] = *(const f32x4 *)(p.epi + 16 * mt + 4 * q) * X3_WSCALE;
                    }
                }
                if (nt < NTT) {
                    xresT[nt] = skip(sw[3], v);
                    accT[nt] = bias_t();
                    accx[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                } else if (EXM) {
                    xrese = skip(swe, v);
                    acce = *(const f32x4 *)(p.epi + 16 * emt + 4 * q) * X3_WSCALE;
                }
            }
        }
        // scratch addresses.  Lane (q, l15) of tile X holds rows 4q..4q+3 = (plane 2q, c0), (2q, c1), (2q+1, c0), (2q+1, c1) of its
        // column; plane t belongs to tap tap_of_plane(t) with d = (dy, dx): the value is a term of out[c, position - d], whose
        // column the host's table names (0xFFFF: off the board -> the trash slot).  Plane 8 = the centre tap (rows 4..7 of tile T).
        int sdst[NTT][2], scen[NTT];
#pragma unroll
        for (int nt = 0; nt < NTT; nt++) {
            const int col = tile[nt] * 16 + l15;
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int t = 2 * q + k, d = p.xd_sdst[col * 8 + t];
                sdst[nt][k] = (grow[nt] >= 0 && d != 0xFFFF) ? X3D::OFF_S + t * S_PLANE + d * 8 : trash;
            }
            scen[nt] = (q == 1 && grow[nt] >= 0) ? X3D::OFF_S + 8 * S_PLANE + col * 8 : trash;
        }

        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads(); // chunks 0, 1 and the parameters have landed; the input planes are written

        int chunk = 0; // the chunk of the current k-step
        auto conv_step = [&](int conv, auto is_first_c) {
            constexpr bool IS_FIRST = decltype(is_first_c)::value;
            constexpr int NKSC = IS_FIRST ? AZ_NET_K0STEPS : NKS;
            using K = X3DK<IS_FIRST>;
            half8 ah[2][3], al[2][3]; // A fragments (weights) of tiles 0..2; k-step ks uses ring slot ks & 1
            half8 at, axh, axl;       // ... of tiles T and X
            half8 bh[NTA], bl[NTA];                          // B fragments (activations) of the CURRENT k-step, per tile
            f32x2 s49[NTT];                                  // (lanes q == 0) sum over the nine tap planes at this lane's column: channels 48, 49
            using SCH = X3DS<IS_FIRST, NTW, EXM, EXT>;
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            unsigned gh[NTA][4], gl[NTA][4]; // the gather k-step's B fragments, dword by dword (assembled at the MFMA, behind the wait)
            auto read_a = [&](unsigned wb, auto ks_c, auto r_c) { // fragment r of k-step ks into ring slot ks & 1: ah 0..2, al 0..2, T, X hi, X lo
                constexpr int ks = decltype(ks_c)::value, r = decltype(r_c)::value, buf = ks & 1;
                constexpr int off = SCH::a_off(ks, r);
                if constexpr (r < 3) READ_A(ah[buf][r], wb, off);
                else if constexpr (r < 6) READ_A(al[buf][r - 3], wb, off);
                else if constexpr (r == 6) READ_A(at, wb, off);
                else if constexpr (r == 7) READ_A(axh, wb, off);
                else READ_A(axl, wb, off);
            };
            unsigned kcur = 0; // this lane's k-group offset for the k-step whose B fragments are being fetched
            auto set_k = [&](auto ks_c) {
                constexpr int ks = decltype(ks_c)::value;
                if constexpr (!K::is_gather(ks)) {
                    const int qq = opaque(q);
                    kcur = (unsigned)sel4(qq, koff_of(ks, 0, IS_FIRST), koff_of(ks, 1, IS_FIRST), koff_of(ks, 2, IS_FIRST), koff_of(ks, 3, IS_FIRST));
                }
            };
            auto read_b = [&](auto ks_c, auto nt_c) { // the B fragments of tile nt for k-step ks (set_k(ks) first; address arithmetic: one or four adds)
                constexpr int ks = decltype(ks_c)::value, nt = decltype(nt_c)::value;
                if constexpr (K::is_gather(ks)) {
                    const int qq = opaque(q);
                    const unsigned p6v = (unsigned)p6_of(nt);
                    static_for<4>([&](auto i_c) {
                        constexpr int i = decltype(i_c)::value;
                        const unsigned a = addr_add(p6v, (unsigned)sel4(qq, ksp_of(i, 0), ksp_of(i, 1), ksp_of(i, 2), ksp_of(i, 3)));
                        READ_B32_OFF(gh[nt][i], a, 0);
                        READ_B32_OFF(gl[nt][i], a, LO_OFF);
                    });
                } else {
                    const unsigned a = addr_add(baseL[nt], kcur);
                    READ_B_OFF(bh[nt], a, 0);
                    READ_B_OFF(bl[nt], a, LO_OFF);
                }
            };
            asm volatile("" ::: "memory"); // (the epilogue's LDS stores stay above these untracked reads)
            { // k-step 0: nothing of it can be fetched before the barrier behind the epilogue that wrote the planes
                const unsigned wb0 = lds_base + (chunk & 1) * CHUNK_S + lane * 16;
                static_for<K::n_a(0)>([&](auto r_c) { read_a(wb0, std::integral_constant<int, 0>{}, r_c); });
                set_k(std::integral_constant<int, 0>{});
                static_for<SCH::ntb(0)>([&](auto nt_c) { read_b(std::integral_constant<int, 0>{}, nt_c); });
            }
            static_for<NKSC>([&](auto ks_c) {
                constexpr int ks = decltype(ks_c)::value;
                constexpr int part = IS_FIRST ? ks / 2 : X3D::part_of(ks);
                constexpr bool last_of_chunk = IS_FIRST ? (ks & 1) : ks == X3D::part_ks0(part) + X3D::part_len(part) - 1;
                constexpr bool last_of_conv = ks == NKSC - 1;
                constexpr int cur = ks & 1;
                constexpr bool T_ON = K::has_t(ks), X_ON = K::has_x(ks);
                constexpr int NTB = SCH::ntb(ks), NTBN = last_of_conv ? 0 : SCH::ntb(ks + 1), N_ANEXT = last_of_conv ? 0 : K::n_a(ks + 1);
                constexpr bool GATHER = K::is_gather(ks);
                const unsigned wb_cur = lds_base + (chunk & 1) * CHUNK_S + lane * 16, wb_oth = lds_base + ((chunk + 1) & 1) * CHUNK_S + lane * 16;
                const unsigned wb_next = last_of_chunk ? wb_oth : wb_cur; // where the next k-step's fragments live
                // the A fragments and tile 0's B fragments of this k-step; at the conv's last k-step every plane read (the epilogue
                // stores follow its barrier), and around the scratch path everything (its LDS operations are not in the count)
                constexpr bool S_STORE = !IS_FIRST && ks == 8, S_SUM = !IS_FIRST && ks == 11;
                wait_lgkm((last_of_conv || S_STORE || S_SUM || K::late_tx(ks)) ? 0 : SCH::wait_n(ks, 0));
                if constexpr (last_of_chunk) { // the chunk's fragments are in registers: its buffer is free, and the next chunk must be visible
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    if (!IS_FIRST && part == 0 && wave == 0) // this conv's epilogue parameters ride the same DMA path into a 2-slot ring
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const unsigned char *)p.epi + (size_t)conv * 1024 + lane * 16),
                                                         (__attribute__((address_space(3))) void *)(lds + X3D::OFF_EPI + (conv & 1) * 1024), 16, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                // channels 48, 49 (az_tower_x3b.h): tile X and the centre-tap rows of tile T are final after k-step 7.  k-step 8: every
                // value goes to its destination column in the workgroup's scratch; the barrier of k-step 10 orders the waves; k-step 11:
                // the lanes of channels 48, 49 sum the nine tap planes at their column.
                if constexpr (S_STORE) {
#pragma unroll
                    for (int nt = 0; nt < NTT; nt++) {
                        const f32x4 xv = accx[nt] * INV_SPLIT;
                        f32x2 cv; // (lanes q == 1 hold rows 4..7 of tile T: hi c0, hi c1, lo c0, lo c1 of the centre tap)
#pragma unroll
                        for (int i = 0; i < 2; i++) cv[i] = (accT[nt][i] + accT[nt][i + 2]) * INV_SPLIT;
                        lds_write64(lds_base + sdst[nt][0], (f32x2){xv[0], xv[1]});
                        lds_write64(lds_base + sdst[nt][1], (f32x2){xv[2], xv[3]});
                        lds_write64(lds_base + scen[nt], cv);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                if constexpr (S_SUM) { // all the tiles' reads first, one wait per tile (counted: the later tiles' reads are still in flight)
                    f32x2 pl[NTT][9];
                    static_for<NTT>([&](auto nt_c) {
                        constexpr int nt = decltype(nt_c)::value;
                        const unsigned sread = lds_base + X3D::OFF_S + (tile[nt] * 16 + l15) * 8;
                        static_for<9>([&](auto t_c) { lds_read64_off<decltype(t_c)::value * S_PLANE>(pl[nt][decltype(t_c)::value], sread); });
                    });
                    static_for<NTT>([&](auto nt_c) {
                        constexpr int nt = decltype(nt_c)::value;
                        wait_lgkm(9 * (NTT - 1 - nt) < 15 ? 9 * (NTT - 1 - nt) : 15);
                        static_for<9>([&](auto t_c) { launder(pl[nt][decltype(t_c)::value]); });
                        f32x2 sm = pl[nt][0];
#pragma unroll
                        for (int t = 1; t < 9; t++) sm = sm + pl[nt][t];
                        s49[nt] = sm;
                    });
                }
                const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
                static_for<NTB>([&](auto nt_c) {
                    constexpr int nt = decltype(nt_c)::value;
                    constexpr bool OWN = nt < NTW;
                    constexpr int n_main = OWN ? 9 : (EXM ? 3 : 0), n_t = (T_ON && nt < NTT) ? 2 : 0, n_x = (X_ON && nt < NTT) ? 3 : 0;
                    constexpr int NMT = n_main + n_t + n_x;
                    constexpr int slot0 = nt < NTW ? nt * (9 + (T_ON ? 2 : 0) + (X_ON ? 3 : 0)) : NTW * (9 + (T_ON ? 2 : 0) + (X_ON ? 3 : 0)); // k-step-wide index of the tile's first MFMA
                    if constexpr (nt > 0) wait_lgkm(SCH::wait_n(ks, nt));
                    half8 b_hi, b_lo; // this tile's B fragments (every read of them has landed: pin the registers behind the wait)
                    if constexpr (GATHER) {
                        static_for<4>([&](auto i_c) {
                            launder_u(gh[nt][decltype(i_c)::value]);
                            launder_u(gl[nt][decltype(i_c)::value]);
                        });
                        b_hi = __builtin_bit_cast(half8, (u32x4){gh[nt][0], gh[nt][1], gh[nt][2], gh[nt][3]});
                        b_lo = __builtin_bit_cast(half8, (u32x4){gl[nt][0], gl[nt][1], gl[nt][2], gl[nt][3]});
                    } else {
                        launder(bh[nt]);
                        launder(bl[nt]);
                        b_hi = bh[nt];
                        b_lo = bl[nt];
                    }
                    if constexpr (nt == 0 && !last_of_conv) set_k(std::integral_constant<int, ks + 1>{}); // (tile 0's own fragments are in b_hi / b_lo by now)
                    static_for<NMT>([&](auto j_c) {
                        constexpr int j = decltype(j_c)::value, gslot = slot0 + j;
                        // fetches for the next k-step: its A fragments one per slot of tile 0 (at most nine: tile 0 has at least nine slots), the B fragments of tile nt - 1 behind tile nt's first MFMA
                        if constexpr (nt == 0 && j < N_ANEXT) read_a(wb_next, std::integral_constant<int, ks + 1>{}, std::integral_constant<int, j>{});
                        if constexpr (nt >= 1 && j == 1 && nt - 1 < NTBN) read_b(std::integral_constant<int, ks + 1>{}, std::integral_constant<int, (nt >= 1 ? nt - 1 : 0)>{});
                        if constexpr (last_of_chunk && gslot >= 2 && (gslot - 2) % 3 == 0 && (gslot - 2) / 3 < NPW) { // buffer chunk & 1 is free: fetch chunk + 2
                            if constexpr (IS_FIRST) // conv 0's part -> the same part of conv 1
                                issue_piece(chunk & 1, (size_t)X3D::C0_B + X3D::part_off(part), std::integral_constant<int, X3D::part_bytes(part) / 1024>{}, (gslot - 2) / 3);
                            else {
                                constexpr int part2 = (part + 2) % PARTS, dconv = (part + 2) / PARTS;
                                issue_piece(chunk & 1, (size_t)X3D::C0_B + (size_t)(conv - 1 + dconv) * X3D::CONV_B + X3D::part_off(part2),
                                            std::integral_constant<int, X3D::part_bytes(part2) / 1024>{}, (gslot - 2) / 3);
                            }
                        }
                        if constexpr (j < n_main) {
                            // per accumulator and k-step: hi'*hi, hi'*lo0, lo*hi (the order of az_tower_x3b_kernel: same bits); within a tile
                            // pass-major, so that an accumulator's three MFMAs are three slots apart
                            constexpr int pass = OWN ? j / 3 : j, mt = OWN ? j % 3 : 0;
                            if constexpr (OWN) {
                                const half8 a_hi = ah[cur][mt], a_lo = al[cur][mt];
                                if constexpr (pass == 0) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, b_hi, acc[mt][nt], 0, 0, 0);
                                else if constexpr (pass == 1) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, b_lo, acc[mt][nt], 0, 0, 0);
                                else acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo, b_hi, acc[mt][nt], 0, 0, 0);
                            } else { // the split tile's output-channel tile mt = emt
                                if constexpr (pass == 0) acce = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[cur][emt], b_hi, acce, 0, 0, 0);
                                else if constexpr (pass == 1) acce = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[cur][emt], b_lo, acce, 0, 0, 0);
                                else acce = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[cur][emt], b_hi, acce, 0, 0, 0);
                            }
                        } else if constexpr (j < n_main + n_t) { // tile T (hi rows and lo rows in one fragment): x B_hi, then x B_lo0
                            if constexpr (j == n_main) accT[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(at, b_hi, accT[nt], 0, 0, 0);
                            else accT[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(at, b_lo, accT[nt], 0, 0, 0);
                        } else { // tile X: hi'*hi (from a literal 0 at k-step 6), hi'*lo0, lo*hi
                            constexpr int jj = j - n_main - n_t;
                            if constexpr (jj == 0) accx[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(axh, b_hi, ks == 6 ? zero4 : accx[nt], 0, 0, 0);
                            else if constexpr (jj == 1) accx[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(axh, b_lo, accx[nt], 0, 0, 0);
                            else accx[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(axl, b_hi, accx[nt], 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    });
                });
                // the B fragments of the next k-step that had no later tile to ride behind; its T and X fragments where this k-step used their registers
                if constexpr (!last_of_conv) {
                    static_for<NTBN>([&](auto nt_c) {
                        constexpr int nt = decltype(nt_c)::value;
                        if constexpr (nt + 1 >= NTB) read_b(std::integral_constant<int, ks + 1>{}, nt_c);
                    });
                    if constexpr (K::late_tx(ks + 1) && NTT > 0)
                        static_for<K::n_tx(ks + 1)>([&](auto r_c) { read_a(wb_next, std::integral_constant<int, ks + 1>{}, std::integral_constant<int, 6 + decltype(r_c)::value>{}); });
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (last_of_chunk) chunk++;
            });
            // ---- epilogue, in fp32; the result is split into (hi, lo) again (az_tower_x3b.h: the same arithmetic) -------------
            const unsigned ep_base = lds_base + X3D::OFF_EPI + (conv & 1) * 1024 + q * 16;
            // sa_hi / sa_lo: LDS address of the lane's quarter-octet (q) of output-channel tile 0 in the tile's cell, hi and lo planes (the
            // output-channel tile is an immediate offset of the store: one address per tile and plane set, not three instructions per store)
            auto unit = [&](auto kind, auto tt_c, f32x4 &a, f32x4 &xr, const f32x2 &s, auto mt_c, const int nt,
                            const f32x4 &sc, const f32x4 &sh, const f32x4 &next_bias, const unsigned sa_hi, const unsigned sa_lo) {
                constexpr int KIND = decltype(kind)::value; // 0: conv1, 1: conv2 (not last), 2: last conv
                constexpr bool TT = decltype(tt_c)::value;  // tile T
                constexpr int mt = decltype(mt_c)::value;
                const int co0 = 16 * mt + 4 * q;
                f32x4 v;
                const f32x4 a_in = a;
                if constexpr (!TT) {
                    v = a * INV_SPLIT;
                    a = next_bias; // (2048 x the next conv's bias: scaled on the host)
                } else { // lanes q == 0: rows hi 48, hi 49, lo 48, lo 49 of the gather k-step (+ bias), plus the tap planes
                    v = (f32x4){(a[0] + a[2]) * INV_SPLIT, (a[1] + a[3]) * INV_SPLIT, 0.f, 0.f};
                    if constexpr (!IS_FIRST) {
                        v[0] += s[0];
                        v[1] += s[1];
                    }
                    if (q != 0) v = (f32x4){0.f, 0.f, 0.f, 0.f}; // (rows 4..15: centre-tap rows / unused)
                    a = q == 0 ? (f32x4){next_bias[0], next_bias[1], 0.f, 0.f} : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
                f32x4 o;
                if constexpr (KIND == 0) {
                    o = __builtin_elementwise_max(v, v * 0.01f);
                } else {
                    // xr + a / 2048 in one instruction: a / 2048 is exact (a power of two), so the fma has the bits of multiply-then-add
                    f32x4 xv;
                    if constexpr (!TT) xv = __builtin_elementwise_fma(a_in, (f32x4){INV_SPLIT, INV_SPLIT, INV_SPLIT, INV_SPLIT}, xr);
                    else xv = xr + v;
                    xr = xv;
                    if constexpr (KIND == 2) {
                        half4 hi, lo;
                        split4(xv, hi, lo);
                        if (((opaque((int)livem) >> nt) & 1) && co0 < p.xout_c) { // a board of the batch (the column's global row, looked up again: once per
                                                                                // forward); channels past the stride (tile T, q > 0) are not stored
                            const int e = p.xd_pos[tile[nt] * 16 + l15];
                            const size_t gr = (size_t)(blockIdx.x * p.xd_nb + (e >> 8)) * p.HW + (e & 255);
                            *(half4 *)(p.xout + gr * p.xout_c + co0) = hi;
                            *(half4 *)(p.xout_lo + gr * p.xout_c + co0) = lo;
                        }
                        return;
                    }
                    f32x4 t = __builtin_elementwise_fma(sc, xv, sh);
                    o = __builtin_elementwise_max(t, t * 0.01f);
                }
                half4 hi, lo;
                split4_planes(o, hi, lo); // (between convs the lo half is unscaled; the tower OUTPUT above keeps lo x 2048: the head's format)
                // Every lane of every tile is a real position of one of the workgroup's boards (the tiles are full), and a board past the
                // batch computes on zero planes into cells nobody reads: no store needs a mask (x3b's padding lanes do: their cell is
                // the halo column).  Tile T: only the lanes that hold channels 48, 49 (q == 0) store.
                if constexpr (TT) { // channels 48, 49 -> the compact planes
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                    if (q == 0) {
                        const int a6 = p6_of(nt) - (int)lds_base;
                        *(unsigned *)(lds + a6) = __builtin_bit_cast(u32x2, hi)[0];
                        *(unsigned *)(lds + a6 + LO_OFF) = __builtin_bit_cast(u32x2, lo)[0];
                    }
                } else {
                    lds_write64_off<2 * mt * plane_b>(sa_hi, __builtin_bit_cast(f32x2, hi));
                    lds_write64_off<2 * mt * plane_b>(sa_lo, __builtin_bit_cast(f32x2, lo));
                }
            };
            auto epilogue = [&](auto kind) {
                // parameters of output-channel tile mt: scale, shift (of the NEXT prologue), bias of the next conv (the ring slot's rows 1..3)
                auto params = [&](const int mt, f32x4 &sc, f32x4 &sh, f32x4 &nb) {
                    if constexpr (!IS_FIRST) {
                        sc = *(const f32x4 *)(lds + (ep_base - lds_base) + 256 + mt * 64);
                        sh = *(const f32x4 *)(lds + (ep_base - lds_base) + 512 + mt * 64);
                    }
                    nb = *(const f32x4 *)(lds + (ep_base - lds_base) + 768 + mt * 64);
                };
                const f32x2 s0 = {0.f, 0.f};
                // every tile's parameters in ONE batch of LDS reads (a read-wait per tile is five exposed round trips per epilogue)
                f32x4 sc[4], sh[4], nb[4];
#pragma unroll
                for (int mt = 0; mt < 4; mt++) {
                    sc[mt] = sh[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    params(mt, sc[mt], sh[mt], nb[mt]);
                }
                // (opaque: the addresses are invariant across convs, and hoisted out of the conv loop they would hold registers for the whole kernel)
                unsigned sa_hi[NTA], sa_lo[NTA];
                const unsigned qoff = (unsigned)((q >> 1) * plane_b + (q & 1) * 8);
#pragma unroll
                for (int nt = 0; nt < NTA; nt++) {
                    sa_hi[nt] = addr_add((unsigned)opaque((int)baseL[nt]), qoff);
                    sa_lo[nt] = sa_hi[nt] + LO_OFF;
                }
                static_for<3>([&](auto mt_c) {
                    constexpr int mt = decltype(mt_c)::value;
#pragma unroll
                    for (int nt = 0; nt < NTW; nt++) unit(kind, std::false_type{}, acc[mt][nt], xres[mt][nt], s0, mt_c, nt, sc[mt], sh[mt], nb[mt], sa_hi[nt], sa_lo[nt]);
                });
#pragma unroll
                for (int nt = 0; nt < NTT; nt++)
                    unit(kind, std::true_type{}, accT[nt], xresT[nt], IS_FIRST ? s0 : s49[nt], std::integral_constant<int, 3>{}, nt, sc[3], sh[3], nb[3], sa_hi[nt], sa_lo[nt]);
                if constexpr (EXM)
                    unit(kind, std::false_type{}, acce, xrese, s0, std::integral_constant<int, EXM ? emt : 0>{}, NTW, sc[emt], sh[emt], nb[emt], sa_hi[NTW], sa_lo[NTW]);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // (the plane stores are untracked asm statements)
            };
            if constexpr (IS_FIRST) epilogue(std::integral_constant<int, 0>{});
            else {
                if (!(conv & 1)) epilogue(std::integral_constant<int, 0>{});
                else if (conv != p.n_convs - 1) epilogue(std::integral_constant<int, 1>{});
                else epilogue(std::integral_constant<int, 2>{});
            }
            __syncthreads(); // every wave's part of the new activations is in the planes before anybody reads them
        };
        conv_step(0, std::true_type{});
        for (int conv = 1; conv < p.n_convs; conv++) conv_step(conv, std::false_type{});
    };
    constexpr auto I0 = std::integral_constant<int, 0>{};
    constexpr auto I1 = std::integral_constant<int, 1>{};
    constexpr auto I2 = std::integral_constant<int, 2>{};
    constexpr auto I3 = std::integral_constant<int, 3>{};
    (void)I3;
    if constexpr (V == 0) {
        if ((wave & 3) == 0) body(I2, I1, I0);
        else if ((wave & 3) == 1) body(I2, I1, I1);
        else if ((wave & 3) == 2) body(I2, I1, I2);
        else body(I2, I2, I0);
    } else if constexpr (V == 2) {
        if (wave < 4) body(I3, I0, I0);
        else if (wave == 4) body(I2, I1, I0);
        else if (wave == 5) body(I2, I1, I1);
        else if (wave == 6) body(I2, I1, I2);
        else body(I2, I2, I0);
    } else body(I2, I0, I0);
}
