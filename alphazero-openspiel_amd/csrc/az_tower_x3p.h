// az_tower_x3p.h — az_tower_x3p_kernel: az_tower_x3b_kernel's arithmetic (fp32-grade, split-fp16 operands, no output-channel
// tile for channels 48, 49) with TWO waves per board, both on one SIMD: the full-batch kernel of the product default since round 4.
// Reference computation: ResidualBlock.forward x n_blocks of Net.forward (network.py:48-64,99-104) in eval mode.
//
// az_tower_x3b_kernel runs one wave per SIMD (a board per wave, 154 KB of LDS per four boards: a second workgroup does not fit).
// A lone wave issues in order, so everything that is not an MFMA is exposed: the fp32 epilogues (~600 vector instructions per conv
// at 4 cycles each when one wave issues them alone), the LDS stores, the k-step and chunk waits - the matrix pipe was busy 52 %
// of the wave's life (profiles/r3_bench_default_pmc_summary.txt).  Here a workgroup is EIGHT waves on the same four boards and the
// same LDS map; wave w and wave w + 4 (a workgroup's waves are dealt to the SIMDs cyclically: they share SIMD w) split board w
// by output-channel tile:
//     role 0 (waves 0..3): tiles mt = 0, 1                       18 MFMAs per k-step, 270 per conv, 6 epilogue units
//     role 1 (waves 4..7): tile mt = 2, tiles T and X (48, 49)    9 (+ 6 T, + 9 X) per k-step, 171 per conv, 4 units + scratch path
// The pair feeds ONE matrix pipe with the same 441 MFMAs per conv as before, but two instruction streams: one wave's waits, LDS
// traffic and epilogue arithmetic sit beside the other's MFMAs, and where both are in vector code the SIMD issues it at two
// cycles per instruction instead of four.  Both waves read every B (activation) fragment and only their own A (weight) fragments.
// The pair shares the board's planes, so (as in az_tower_x3c_kernel):
//   * write-after-read: epilogue stores follow the barrier of the conv's last k-step, and every wave's last plane read (the gather
//     k-step's B fragments) has returned before it enters that barrier;
//   * read-after-write: one more barrier per conv, between the epilogue stores and the next conv's first B reads.
// Every accumulator sees the same MFMAs in the same order as in az_tower_x3b_kernel / az_tower_x3c_kernel and the epilogue
// arithmetic is the same code: a board's outputs are the same BITS whichever of the three kernels evaluates it.
#pragma once
#include "az_tower_x3b.h"

// what a wave of role ROLE multiplies and reads in k-step ks
template <bool IS_FIRST, int NT, int ROLE> struct X3PK {
    using K = X3BK<IS_FIRST, NT>;
    static constexpr int MT0 = ROLE == 0 ? 0 : 2, NMT = ROLE == 0 ? 2 : 1;
    static constexpr bool TX = ROLE == 1;
    static constexpr bool has_t(int ks) { return TX && K::has_t(ks); }
    static constexpr bool has_x(int ks) { return TX && K::has_x(ks); }
    static constexpr int n_a(int ks) { return 2 * NMT; } // prefetched a k-step ahead: ah, al of its tiles (T, X: read at their k-step's top)
    static constexpr int n_b(int ks) { return K::n_b(ks); }
    static constexpr int n_mfma(int ks) { return 3 * NMT * NT + (has_t(ks) ? 2 * NT : 0) + (has_x(ks) ? 3 * NT : 0); }
};

template <int NT>
__global__ __launch_bounds__(512, 2) void az_tower_x3p_kernel(TowerParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int WAVES = 8, FR = X3B::FR, REC2 = X3B::REC2, CK = X3B::CK, NKS = X3B::NKS, PARTS = X3B::PARTS;
    constexpr int CHUNK_S = X3B::CHUNK_S, LO_OFF = X3B::LO_OFF, S_PLANE = X3B::S_PLANE;
    constexpr float INV_SPLIT = 1.0f / 2048.0f, SPLIT = 2048.0f;
    constexpr int plane_b = X3B::PLANE_B;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, l15 = lane & 15;
    const int bl = wave & 3, role = wave >> 2, tid_b = role * 64 + lane; // board of the workgroup, role in the pair, thread in the pair
    const int board0 = blockIdx.x * 4 + bl;
    const int region = X3B::OFF_ACT + bl * 2 * LO_OFF; // (x3b's map: planes and scratch of its wave bl)
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds;
    const int trash = X3B::OFF_EPI + 2048 + (tid & 255) * 16; // dump slot for masked-out stores (the pair's waves share them: write-only)
    const int s_wave = X3B::OFF_S + bl * X3B::S_WAVE;

    { // zero the planes and the scratch, the pair together
        uint4 z = {0, 0, 0, 0};
        for (int i = tid_b * 16; i < 2 * LO_OFF; i += 128 * 16) *(uint4 *)(lds + region + i) = z;
        for (int i = tid_b * 16; i < X3B::S_WAVE; i += 128 * 16) *(uint4 *)(lds + s_wave + i) = z;
    }
    TowerTables<NT, true, true> T;
    T.init(p, region, plane_b, lds_base, board0, q, l15);
    int (&pos_addr)[NT] = T.pos_addr, (&grow)[NT] = T.grow, (&p6_addr)[NT] = T.p6_addr;
    int (&koff)[AZ_NET_KSTEPS] = T.koff, (&ksp)[4] = T.ksp, (&koff0)[AZ_NET_K0STEPS] = T.koff0;

    auto split4 = [&](const f32x4 &v, half4 &hi, half4 &lo) {
        hi = __builtin_convertvector(v, half4);
        lo = __builtin_convertvector((v - __builtin_convertvector(hi, f32x4)) * SPLIT, half4);
    };
    __syncthreads(); // the zeroes are down before role 0 writes the input planes

    // ---- weight stream (az_tower_x3b.h): chunk c -> buffer c & 1 by LDS-DMA, a KiB per wave-instruction; eight waves share the pieces
    auto issue_piece = [&](int c, auto part_c, int i) { // piece 8 i + wave of chunk c (= part `part` of its conv; part -1: conv 0)
        constexpr int part = decltype(part_c)::value;
        constexpr int NPIECES = (part < 0 ? X3B::C0_B : X3B::part_bytes(part < 0 ? 0 : part)) / 1024;
        int piece = i * WAVES + wave; // (branch-free: past the chunk -> its last piece again; past the stream -> zero padding)
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

    auto body = [&](auto role_c) {
        constexpr int ROLE = decltype(role_c)::value;
        constexpr int MT0 = ROLE == 0 ? 0 : 2, NMT = ROLE == 0 ? 2 : 1;
        constexpr bool TX = ROLE == 1;
        constexpr int NA = NMT + (TX ? 1 : 0); // accumulator sets: the wave's tiles, then tile T
        auto tile_of = [](int li) constexpr { return li < NMT ? MT0 + li : 3; }; // index in the parameter tables (T = 3)
        f32x4 acc[NA][NT], acc2[NA][NT], xres[NA][NT];
        f32x4 accxh[NT], accxl[NT]; // ROLE 1: tile X
        { // prologue: role 0 writes a = lrelu(bn1(x0)) -> octet 0 (hi, lo); every wave takes its tiles' share of the block-1 skip conv
            f32x4 sw[NA][4];
#pragma unroll
            for (int li = 0; li < NA; li++)
#pragma unroll
                for (int r = 0; r < 4; r++) sw[li][r] = *(const f32x4 *)(p.skip_w + (16 * tile_of(li) + 4 * q + r) * 4);
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
                        split4(a, hi, lo);
                        *(half4 *)(lds + pos_addr[nt]) = hi;
                        *(half4 *)(lds + pos_addr[nt] + LO_OFF) = lo;
                    }
                }
#pragma unroll
                for (int li = 0; li < NA; li++) {
                    const int mt = tile_of(li);
                    f32x4 x;
#pragma unroll
                    for (int r = 0; r < 4; r++) x[r] = sw[li][r][0] * v[0] + sw[li][r][1] * v[1] + sw[li][r][2] * v[2] + sw[li][r][3] * v[3];
                    xres[li][nt] = x;
                    // tile T: only channels 48, 49 (lanes q == 0, rows 0, 1) carry a bias; its other rows are lo / centre-tap rows
                    acc[li][nt] = (mt < 3 || q == 0) ? *(const f32x4 *)(p.epi + 16 * mt + 4 * q) : (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (mt == 3) acc[li][nt][2] = acc[li][nt][3] = 0.f;
                    acc2[li][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
                accxh[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                accxl[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        // scratch addresses (ROLE 1; az_tower_x3b.h)
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

        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads(); // chunks 0, 1 and the parameters have landed; the input planes are written

        int chunk = 0;
        auto conv_step = [&](int conv, const auto &kf, auto is_first_c) {
            constexpr bool IS_FIRST = decltype(is_first_c)::value;
            constexpr int NKSC = IS_FIRST ? AZ_NET_K0STEPS : NKS;
            using K = X3BK<IS_FIRST, NT>;
            using R = X3PK<IS_FIRST, NT, ROLE>;
            half8 ah[2][NMT], al[2][NMT]; // A fragments of the wave's tiles: k-step ks uses ring slot ks & 1
            half8 at, axh, axl;           // tiles T and X (three k-steps of a conv): read at the top of their k-step - a ring for them is
                                          // 12 registers this wave does not have; it has a third of the pair's MFMAs and can wait
            half8 bh[2][NT], bl[2][NT];                          // B fragments (activations), hi / lo
            unsigned sph[NT][4], spl[NT][4];                     // gather k-step: B fragments dword by dword
            f32x4 ep_sc[NA], ep_sh[NA], ep_nb[NA];
            const unsigned ep_base = lds_base + X3B::OFF_EPI + (conv & 1) * 1024 + q * 16;
            // fragment read r of k-step ks into ring slot buf.  Order: ah of the wave's tiles, al of them, T, X hi, X lo
            auto read_a = [&](unsigned wb, auto buf_c, auto ks_c, auto r_c) {
                constexpr int buf = decltype(buf_c)::value, ks = decltype(ks_c)::value, r = decltype(r_c)::value;
                constexpr int part = IS_FIRST ? 0 : ks / CK, ksl = ks - part * CK;
                if constexpr (r < NMT) READ_A(ah[buf][r], wb, ksl * REC2 + (MT0 + r) * FR);
                else READ_A(al[buf][r - NMT], wb, ksl * REC2 + (3 + MT0 + r - NMT) * FR);
            };
            auto read_b = [&](auto buf_c, auto ks_c, auto r_c) { // plain: hi 0..NT-1, lo 0..NT-1; gather: tile-major dwords
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
            f32x2 s49[NT]; // (ROLE 1, lanes q == 0) sum over the nine tap planes at this lane's position: channels 48, 49
            asm volatile("" ::: "memory"); // (the epilogue's LDS stores stay above these untracked reads)
            // k-step 0: its B fragments cannot be fetched before the barrier behind the epilogue that wrote them; its A fragments ride
            // in the same wait (x3b fetches them a k-step early into registers of their own: with two waves per SIMD the partner covers
            // the latency, and the 16 registers are what lets two waves fit).  The chunk they sit in landed before the barrier of the
            // previous conv's last k-step.
            static_for<R::n_b(0)>([&](auto r_c) { read_b(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, r_c); });
            {
                const unsigned wb0 = lds_base + (chunk & 1) * CHUNK_S + lane * 16;
                static_for<R::n_a(0)>([&](auto r_c) { read_a(wb0, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, r_c); });
            }
            static_for<NKSC>([&](auto ks_c) {
                constexpr int ks = decltype(ks_c)::value;
                constexpr int part = IS_FIRST ? 0 : ks / CK, ksl = ks - part * CK;
                constexpr int CKL = IS_FIRST ? AZ_NET_K0STEPS : (part == PARTS - 1 ? NKS - part * CK : CK);
                constexpr bool last_of_chunk = ksl == CKL - 1, last_of_conv = ks == NKSC - 1;
                constexpr int cur = ks & 1, nxt = cur ^ 1;
                constexpr int na_next = last_of_conv ? 0 : R::n_a(ks + 1);
                constexpr int n_next = last_of_conv ? 0 : na_next + R::n_b(ks + 1);
                constexpr bool T_ON = R::has_t(ks), X_ON = R::has_x(ks), GATHER = K::is_gather(ks);
                constexpr int NM = R::n_mfma(ks), NMAIN = 3 * NMT * NT;
                constexpr int RPS = (n_next + NM - 1) / NM > 1 ? (n_next + NM - 1) / NM : 1; // reads of the next k-step per MFMA slot
                constexpr int part2 = IS_FIRST ? 1 : (part + 2) % PARTS;                      // the part chunk + 2 is
                constexpr int PSTEP = NM >= 3 * NPW + 1 ? 3 : 2;                              // a DMA piece every third (second) MFMA slot
                static_assert(1 + PSTEP * (NPW - 1) < NM, "the pieces of chunk + 2 fit the k-step");
                const unsigned wb_cur = lds_base + (chunk & 1) * CHUNK_S + lane * 16, wb_oth = lds_base + ((chunk + 1) & 1) * CHUNK_S + lane * 16;
                const unsigned wb_next = last_of_chunk ? wb_oth : wb_cur; // where the next k-step's fragments live
                if constexpr (T_ON) { // the extra fragments sit behind the chunk's records; read before the barrier that frees the buffer
                    constexpr int xbase = IS_FIRST ? AZ_NET_K0STEPS * REC2 + ks * FR : (part == 1 ? 4 * REC2 + (ks - 6) * 3 * FR : 3 * REC2);
                    READ_A(at, wb_cur, xbase);
                    if constexpr (X_ON) {
                        READ_A(axh, wb_cur, xbase + FR);
                        READ_A(axl, wb_cur, xbase + 2 * FR);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // every other fragment of this k-step was issued early in the previous one
                if constexpr (last_of_chunk) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    if (!IS_FIRST && part == 0 && wave == 0) // this conv's epilogue parameters ride the same DMA path into a 2-slot ring
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const unsigned char *)p.epi + (size_t)conv * 1024 + lane * 16),
                                                         (__attribute__((address_space(3))) void *)(lds + X3B::OFF_EPI + (conv & 1) * 1024), 16, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                static_for<NM>([&](auto j_c) {
                    constexpr int j = decltype(j_c)::value;
                    static_for<RPS>([&](auto rr_c) { // reads of the next k-step, in its read order
                        constexpr int r = RPS * j + decltype(rr_c)::value;
                        if constexpr (r < n_next) {
                            if constexpr (r < na_next)
                                read_a(wb_next, std::integral_constant<int, nxt>{}, std::integral_constant<int, ks + 1>{}, std::integral_constant<int, r>{});
                            else read_b(std::integral_constant<int, nxt>{}, std::integral_constant<int, ks + 1>{}, std::integral_constant<int, r - na_next>{});
                        }
                    });
                    if constexpr (last_of_chunk && j >= 1 && (j - 1) % PSTEP == 0 && (j - 1) / PSTEP < NPW) // buffer chunk & 1 is free: fetch chunk + 2
                        issue_piece(chunk + 2, std::integral_constant<int, part2>{}, (j - 1) / PSTEP);
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
                    if constexpr (j < NMAIN) { // the wave's tiles: pass 0 hi*hi -> acc, pass 1 hi*lo, pass 2 lo*hi -> acc2 (scaled by 2048)
                        constexpr int pass = j / (NMT * NT), nt = (j % (NMT * NT)) / NMT, li = j % NMT;
                        constexpr auto ntc = std::integral_constant<int, nt>{};
                        const half8 a_hi = ah[cur][li], a_lo = al[cur][li];
                        if constexpr (pass == 0) acc[li][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, b_hi(ntc), acc[li][nt], 0, 0, 0);
                        else if constexpr (pass == 1) // (the conv's first product into acc2 starts from a literal 0)
                            acc2[li][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, b_lo(ntc), ks == 0 ? zero4 : acc2[li][nt], 0, 0, 0);
                        else acc2[li][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo, b_hi(ntc), acc2[li][nt], 0, 0, 0);
                    } else if constexpr (T_ON && j < NMAIN + 2 * NT) { // tile T: x B_hi -> acc (hi rows: hi*hi, lo rows: lo*hi), x B_lo -> acc2 (hi rows: hi*lo)
                        constexpr int jj = j - NMAIN, nt = jj % NT;
                        constexpr auto ntc = std::integral_constant<int, nt>{};
                        const half8 a_t = at;
                        if constexpr (jj < NT) acc[NMT][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_t, b_hi(ntc), acc[NMT][nt], 0, 0, 0);
                        else acc2[NMT][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_t, b_lo(ntc), ks == (IS_FIRST ? 0 : 6) ? zero4 : acc2[NMT][nt], 0, 0, 0);
                    } else { // tile X: hi*hi -> accxh; hi*lo, lo*hi -> accxl
                        constexpr int jj = j - NMAIN - 2 * NT, nt = jj % NT;
                        constexpr auto ntc = std::integral_constant<int, nt>{};
                        if constexpr (jj < NT) accxh[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(axh, b_hi(ntc), ks == 6 ? zero4 : accxh[nt], 0, 0, 0);
                        else if constexpr (jj < 2 * NT) accxl[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(axh, b_lo(ntc), ks == 6 ? zero4 : accxl[nt], 0, 0, 0);
                        else accxl[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(axl, b_hi(ntc), accxl[nt], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
                // channels 48, 49: tile X and the centre-tap rows of T are final after k-step 7.  Their shifted sum through the scratch
                // runs here in one piece, behind k-step 8's MFMAs (the same arithmetic, in the same order, as x3b's interleaved
                // version and x3c's): this wave has a third of the pair's MFMAs, and the partner's keep the pipe busy meanwhile
                if constexpr (TX && !IS_FIRST && ks == 8) {
#pragma unroll
                    for (int nt = 0; nt < NT; nt++) {
                        f32x4 xv;
#pragma unroll
                        for (int i = 0; i < 4; i++) xv[i] = accxh[nt][i] + accxl[nt][i] * INV_SPLIT;
                        f32x2 cv;
#pragma unroll
                        for (int i = 0; i < 2; i++) cv[i] = acc[NMT][nt][i] + (acc[NMT][nt][i + 2] + acc2[NMT][nt][i]) * INV_SPLIT;
                        lds_write64(lds_base + sdst[nt][0], (f32x2){xv[0], xv[1]});
                        lds_write64(lds_base + sdst[nt][1], (f32x2){xv[2], xv[3]});
                        lds_write64(lds_base + scen[nt], cv);
                    }
                    // (the reads of k-step 9's fragments, issued above, are still in flight: LDS returns a wave's reads in order, and the
                    //  lgkmcnt(0) below covers them too)
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
            // this conv's epilogue parameters (landed with the barrier of chunk part 1 at the latest), read here: the partner covers the wait
            static_for<NA>([&](auto li_c) {
                constexpr int li = decltype(li_c)::value, mt = li < NMT ? MT0 + li : 3;
                if constexpr (!IS_FIRST) {
                    lds_read_f4_off<256 + mt * 64>(ep_sc[li], ep_base);
                    lds_read_f4_off<512 + mt * 64>(ep_sh[li], ep_base);
                }
                lds_read_f4_off<768 + mt * 64>(ep_nb[li], ep_base);
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            static_for<NA>([&](auto li_c) {
                constexpr int li = decltype(li_c)::value;
                if constexpr (!IS_FIRST) {
                    launder(ep_sc[li]);
                    launder(ep_sh[li]);
                }
                launder(ep_nb[li]);
            });
            // ---- epilogue of this wave's tiles, in fp32; the result is split into (hi, lo) again (az_tower_x3b.h, same arithmetic)
            auto epilogue = [&](auto kind) {
                constexpr int KIND = decltype(kind)::value; // 0: conv1, 1: conv2 (not last), 2: last conv
#pragma unroll
                for (int li = 0; li < NA; li++) {
                    const int mt = tile_of(li);
                    const int co0 = 16 * mt + 4 * q;
                    const int woff = (2 * mt + (q >> 1)) * plane_b + (q & 1) * 8;
                    const f32x4 sc = ep_sc[li], sh = ep_sh[li], next_bias = ep_nb[li];
#pragma unroll
                    for (int nt = 0; nt < NT; nt++) {
                        f32x4 v;
                        if (mt < 3) {
                            v = acc[li][nt] + acc2[li][nt] * INV_SPLIT;
                            acc[li][nt] = next_bias;
                        } else { // tile T, lanes q == 0: rows hi 48, hi 49, lo 48, lo 49 of the gather k-step (+ bias), plus the tap planes
                            v = (f32x4){acc[li][nt][0] + (acc[li][nt][2] + acc2[li][nt][0]) * INV_SPLIT,
                                        acc[li][nt][1] + (acc[li][nt][3] + acc2[li][nt][1]) * INV_SPLIT, 0.f, 0.f};
                            if constexpr (!IS_FIRST) {
                                v[0] += s49[nt][0];
                                v[1] += s49[nt][1];
                            }
                            if (q != 0) v = (f32x4){0.f, 0.f, 0.f, 0.f}; // (rows 4..15: centre-tap rows / unused)
                            acc[li][nt] = q == 0 ? (f32x4){next_bias[0], next_bias[1], 0.f, 0.f} : (f32x4){0.f, 0.f, 0.f, 0.f};
                        }
                        f32x4 o;
                        if (KIND == 0) {
                            o = __builtin_elementwise_max(v, v * 0.01f);
                        } else {
                            f32x4 xv = xres[li][nt] + v;
                            xres[li][nt] = xv;
                            if (KIND == 2) {
                                half4 hi, lo;
                                split4(xv, hi, lo);
                                if (grow[nt] >= 0) {
                                    *(half4 *)(p.xout + (size_t)grow[nt] * AZ_NET_XOUT_C + co0) = hi;
                                    *(half4 *)(p.xout_lo + (size_t)grow[nt] * AZ_NET_XOUT_C + co0) = lo;
                                }
                                continue;
                            }
                            f32x4 a = __builtin_elementwise_fma(sc, xv, sh);
                            o = __builtin_elementwise_max(a, a * 0.01f);
                        }
                        half4 hi, lo;
                        split4(o, hi, lo);
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
            __syncthreads(); // the partner's tiles of the new activations are in the planes before anybody reads them
        };
        conv_step(0, koff0, std::true_type{});
        for (int conv = 1; conv < p.n_convs; conv++) conv_step(conv, koff, std::false_type{});
    };
    if (role == 0) body(std::integral_constant<int, 0>{});
    else body(std::integral_constant<int, 1>{});
}
