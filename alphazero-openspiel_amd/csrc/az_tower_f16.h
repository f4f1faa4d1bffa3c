// az_tower_f16.h — az_tower_kernel: the whole residual tower in one kernel, fp16 MFMA operands (AZ_NET_PREC_F16).
// Reference computation: ResidualBlock.forward x n_blocks of Net.forward (network.py:48-64,99-104) in eval mode.
#pragma once
#include "az_net_common.h"

// LDS image of one wave: 7 channel-octet planes [rcells][8 fp16], plane stride a multiple of 256 B;
// cell = board*cells + (y+1)*rs + (x+1), halo cells stay zero.  Bank behaviour of the B-fragment ds_read_b128: a lane
// group is 8 columns of octet c + 8 columns of octet c+1, so it is conflict-free iff the 16 columns of a tile sit in
// 16 cells that are distinct mod 16.  With a halo column 16 consecutive positions span >= 17 cells (measured: 42 % of
// LDS cycles were conflicts), so for W <= 7 a column tile is TWO WHOLE ROWS at row stride 8: lane l15 -> row 2t + (l15>>3),
// x = l15 & 7 = 16 consecutive cells (x = 7 is the shared halo column: a padding lane).  connect_four: 6 tiles per
// 2 boards either way.  Wider boards keep the generic packing (n = 16*nt + l15 over positions, 2-way conflicts).
// RP1: row-pair tiles with one board per wave: column tile nt sits exactly nt * 256 bytes after tile 0, so a B-fragment
// address is one precomputed register per k-step plus an immediate.
// R3: rows of output-channel tile 3 (channels 48..63) that are stored.  With <= 50 filters only 2 of its 16 rows are real:
// the weight stream then carries, per k-step, three full fragments + 4 x (2 rows + 1 zero row) x 16 B = 3264 B instead
// of 4096 B (-20 % LDS-DMA traffic, the most expensive ingredient of the k-loop); lanes of the missing rows read the
// zero row (same address: a broadcast).  R3 = 16: plain 4 KiB records.
template <int R3> struct WRec {
    static constexpr int ROWS = R3 < 16 ? R3 + 1 : 16;   // stored rows per lane group (incl. the zero row)
    static constexpr int BYTES = 3 * 1024 + 4 * ROWS * 16; // one k-step of weights
};
template <int NT, int CK, int WAVES, bool RP1, int R3>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void az_tower_kernel(TowerParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int REC = WRec<R3>::BYTES;
    constexpr int CHUNK_B = CK * REC;        // bytes of one (full) chunk in the stream
    constexpr int CHUNK_S = CK * 4 * 64 * 16; // stride of the two chunk buffers in LDS (the host's layout)
    // <= 50 filters (R3 < 16): K is grouped into 15 k-steps instead of 16.  Groups 0..53 = (tap, channel octet 0..5);
    // groups 54, 55 zero; the last k-step takes channels 48, 49 of all nine taps: element j of group q < 3 is channel
    // 48 + (j & 1) at tap 4 q + j / 2 - its B fragment is four 4-byte reads (one per tap) instead of one 16-byte read.
    constexpr bool L15 = R3 < 16;
    constexpr int NKS = L15 ? 15 : AZ_NET_KSTEPS;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, l15 = lane & 15;
    const int plane_b = p.rcells * OCT_B, region_b = N_OCT * plane_b;
    const int board0 = (blockIdx.x * WAVES + wave) * p.bpw; // first global board of this wave
    const int region = p.off_act + wave * region_b;
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds;
    const int trash = p.off_epi + 2048 + tid * 8; // per-thread dump slot for masked-out epilogue stores

    { // zero the wave's private planes (halo + padding must read as 0)
        uint4 z = {0, 0, 0, 0};
        for (int i = lane * 16; i < region_b; i += 64 * 16) *(uint4 *)(lds + region + i) = z;
    }
    // ---- per-lane tables: the NT*16 columns of this wave, the k-group offsets of every k-step (az_net_common.h) -----------
    TowerTables<NT, RP1, L15> T;
    T.init(p, region, plane_b, lds_base, board0, q, l15);
    int (&pos_addr)[NT] = T.pos_addr, (&grow)[NT] = T.grow, (&p6_addr)[NT] = T.p6_addr;
    int (&koff)[AZ_NET_KSTEPS] = T.koff, (&ksp)[4] = T.ksp, (&koff0)[AZ_NET_K0STEPS] = T.koff0;

    f32x4 acc[4][NT], xres[4][NT];
    // ---- prologue: a = lrelu(bn1(x0)) -> octet 0; block-1 skip conv3(x0) in fp32 -> residual stream --------
    {
        f32x4 sw[4][4]; // skip weights of this lane's 16 output channels: [mt][r] -> 4 input planes
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
                    half4 a4;
#pragma unroll
                    for (int c = 0; c < 4; c++) a4[c] = c < p.cin ? (_Float16)lrelu(p.in_scale[c] * v[c] + p.in_shift[c]) : (_Float16)0;
                    *(half4 *)(lds + pos_addr[nt]) = a4;
                }
            }
#pragma unroll
            for (int mt = 0; mt < 4; mt++) {
                f32x4 x;
#pragma unroll
                for (int r = 0; r < 4; r++)
                    x[r] = sw[mt][r][0] * v[0] + sw[mt][r][1] * v[1] + sw[mt][r][2] * v[2] + sw[mt][r][3] * v[3];
                xres[mt][nt] = x;
                acc[mt][nt] = *(const f32x4 *)(p.epi + 16 * mt + 4 * q); // bias of conv 0
            }
        }
    }

    // ---- weight stream: chunk c -> buffer c&1, by LDS-DMA (global_load_lds, 16 B/lane).  Chunk 0 = conv 0 (4 k-steps,
    // 16 KiB), chunk c >= 1 = CK k-steps of the 16-k-step convs that follow, contiguous in the device buffer.
    constexpr int PARTS = (NKS + CK - 1) / CK; // chunks per conv (the last one is shorter when NKS = 15)
    constexpr int C0_B = AZ_NET_K0STEPS * REC;
    static_assert(CK % 2 == 0 && AZ_NET_K0STEPS % 2 == 0, "fragment buffer parity relies on an even chunk length");
    static_assert((PARTS & (PARTS - 1)) == 0 && NKS - (PARTS - 1) * CK >= 3, "chunk index arithmetic / the last two k-steps share a chunk");
    static_assert(CK * 4 * 1024 <= 65536, "A-fragment offsets (relative to the chunk base) must fit the ds offset field");
    static_assert(C0_B <= CHUNK_B && REC % 16 == 0 && ((CHUNK_B + 1023) & ~1023) <= CHUNK_S, "chunk must fit its LDS buffer");
    const int n_chunks = 1 + (p.n_convs - 1) * PARTS;
    auto issue_bytes = [&](const unsigned char *src, unsigned char *dst, auto bytes_c) {
        constexpr int NPIECES = (decltype(bytes_c)::value + 1023) / 1024; // the last piece may run past the chunk: the
                                                                          // stream is padded, the LDS buffer has the room
#pragma unroll
        for (int i = 0; i < (NPIECES + WAVES - 1) / WAVES; i++) {
            int piece = i * WAVES + wave; // one KiB per wave-instruction, lane-linear
            if (piece < NPIECES)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + piece * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void *)(dst + piece * 1024), 16, 0, 0);
        }
    };
    auto issue_chunk = [&](int c) { // c >= 1: chunk (c - 1) % PARTS of conv 1 + (c - 1) / PARTS.  A short last chunk is fetched at full
                                    // length (it runs into the next conv's records; the stream is padded at its end)
        const int ci = (c - 1) / PARTS, part = (c - 1) & (PARTS - 1);
        issue_bytes((const unsigned char *)p.conv_w + C0_B + ((size_t)ci * NKS + (size_t)part * CK) * REC, lds + (c & 1) * CHUNK_S,
                    std::integral_constant<int, CHUNK_B>{});
    };
    issue_bytes((const unsigned char *)p.conv_w, lds, std::integral_constant<int, C0_B>{});
    if (wave == 0) // conv 0 has a single chunk: its epilogue parameters must land before that chunk's barrier
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const unsigned char *)p.epi + lane * 16),
                                         (__attribute__((address_space(3))) void *)(lds + p.off_epi), 16, 0, 0);

    int chunk = 0;
    // one conv = NPARTS chunks of CKL k-steps (kf: this lane's k-group offsets) + its epilogue
    auto conv_step = [&](int conv, const auto &kf, auto is_first_c) {
        constexpr bool IS_FIRST = decltype(is_first_c)::value;
        constexpr int NPARTS = IS_FIRST ? 1 : PARTS;
        constexpr int NKSC = IS_FIRST ? AZ_NET_K0STEPS : NKS;      // k-steps of this conv
        constexpr bool HAS_SPECIAL = L15 && !IS_FIRST;             // its last k-step is the 4-byte-gather one
        half8 a[2][4], b[2][NT]; // fragment double buffer: k-step s+1 is fetched while s is multiplied
        unsigned bsp[NT][4];     // B fragments of the gather k-step, dword by dword
        // This conv's epilogue parameters for the lane's 4 x 4 channels (scale, shift, next conv's bias): fetched from the
        // ring during the LAST k-step, when the other fragment buffer is dead, so the epilogue never waits on LDS.
        f32x4 ep_sc[4], ep_sh[4], ep_nb[4];
        const unsigned ep_base = lds_base + p.off_epi + (conv & 1) * 1024 + q * 16;
        static_for<NPARTS>([&](auto part_c) {
            constexpr int part = decltype(part_c)::value;
            constexpr int CKL = part == NPARTS - 1 ? NKSC - part * CK : CK; // k-steps in this chunk
            // The weight fragments are read by untracked asm, so hipcc sees no consumer of the LDS-DMA and would NOT wait
            // for it: wait by hand.  After the barrier every wave's pieces of this chunk have landed and the other
            // buffer is free for the next chunk's DMA.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (chunk + 1 < n_chunks) issue_chunk(chunk + 1);
            if (!IS_FIRST && part == 0 && wave == 0) // this conv's epilogue parameters ride the same DMA path into a 2-slot ring;
                                                     // they land before the next chunk barrier, long before the epilogue reads them
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const unsigned char *)p.epi + (size_t)conv * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void *)(lds + p.off_epi + (conv & 1) * 1024), 16, 0, 0);
            // Fragment reads are inline asm so that the compiler neither sinks them next to their first use nor
            // inserts its own lgkmcnt(0) (left alone it waits every 8 MFMAs: 34 % MFMA utilisation).  Order:
            //   wait(all of k-step ksl) ; for each read of k-step ksl+1: {ds_read ; MFMA of ksl} ; remaining MFMAs
            // so a read has most of an MFMA block (16 cycles per MFMA) to land before the next wait.  All loop indices
            // are compile-time (static_for), so fragment offsets sit in the instructions' offset fields.
            const unsigned wbl = lds_base + (chunk & 1) * CHUNK_S + lane * 16;
            // tile 3: lane (q, l15) reads stored row min(l15, ROWS - 1) of its k-group (the last stored row is zero)
            const unsigned wbl3 = R3 == 16 ? wbl
                                           : lds_base + (chunk & 1) * CHUNK_S +
                                                 (q * WRec<R3>::ROWS + (l15 < WRec<R3>::ROWS - 1 ? l15 : WRec<R3>::ROWS - 1)) * 16;
            // B fragment of column tile nt in k-step ks (ks compile-time, index into kf)
            auto read_b = [&](half8 &dst, auto ks_c, auto nt_c) {
                constexpr int ks = decltype(ks_c)::value, nt = decltype(nt_c)::value;
                if constexpr (RP1) READ_B_OFF(dst, (unsigned)kf[ks], nt * 256);
                else READ_B(dst, lds_base + pos_addr[nt] + opaque(kf[ks])); // opaque: keep the NT*16 sums out of LICM's hands
            };
            // dword i (tap 4q + i, channels 48, 49) of the gather k-step's B fragment for column tile nt
            auto read_bsp = [&](unsigned &dst, auto i_c, auto nt_c) {
                constexpr int i = decltype(i_c)::value, nt = decltype(nt_c)::value;
                if constexpr (RP1) READ_B32_OFF(dst, (unsigned)ksp[i], nt * 64);
                else READ_B32_OFF(dst, lds_base + p6_addr[nt] + opaque(ksp[i]), 0);
            };
            // Read order inside a k-step: A0..A3, B0, B1, ... (read index: A_mt = mt, B_nt = 4 + nt; gather k-step: the four
            // dwords of B_nt are reads 4 + 4 nt .. 7 + 4 nt).  LDS returns in order, so before the MFMAs of column tile nt it
            // is enough to wait until at most (reads issued after B_nt) are outstanding: counted s_waitcnt, not lgkmcnt(0).
            static_for<4>([&](auto mt_c) {
                constexpr int mt = decltype(mt_c)::value;
                READ_A(a[0][mt], mt < 3 ? wbl : wbl3, mt * 1024);
            });
            if constexpr (part == 0) // later chunks of a conv had their B fragments fetched before the barrier
                static_for<NT>([&](auto nt_c) { read_b(b[0][decltype(nt_c)::value], std::integral_constant<int, 0>{}, nt_c); });
            static_for<CKL>([&](auto ksl_c) {
                constexpr int ksl = decltype(ksl_c)::value, ksg = part * CK + ksl; // k-step in the chunk / in the conv
                constexpr int cur = ksl & 1, nxt = cur ^ 1;
                constexpr bool more_here = ksl + 1 < CKL;                   // next k-step is in this chunk: A and B
                constexpr bool more_next = !more_here && part + 1 < NPARTS; // next k-step is in the next chunk: B only
                constexpr bool cur_gather = HAS_SPECIAL && ksg == NKSC - 1;  // this k-step multiplies the gathered fragments
                constexpr bool next_gather = HAS_SPECIAL && ksg + 1 == NKSC - 1; // ... the next one does (same chunk)
                constexpr int n_next = more_here ? (next_gather ? 4 + 4 * NT : NT + 4) : (more_next ? NT : 0); // reads to issue now
                constexpr int RPS = (n_next + 4 * NT - 1) / (4 * NT) > 1 ? (n_next + 4 * NT - 1) / (4 * NT) : 1; // per MFMA slot
                constexpr bool first_of_chunk = ksl == 0;
                constexpr int ks_next = (more_here || more_next) ? ksg + 1 : 0;
                constexpr bool last_of_conv = !more_here && !more_next;
                // (conv 0 is always a conv1-type epilogue: next bias only.  An asynchronous read into a register nothing
                // consumes would let the compiler hand that register to something else while the data is still in flight.)
                constexpr int n_ep = last_of_conv ? (IS_FIRST ? 4 : 12) : 0; // younger reads the counted waits below must allow
                if constexpr (last_of_conv)
                    static_for<4>([&](auto mt_c) {
                        constexpr int mt = decltype(mt_c)::value;
                        if constexpr (!IS_FIRST) {
                            lds_read_f4_off<256 + mt * 64>(ep_sc[mt], ep_base);
                            lds_read_f4_off<512 + mt * 64>(ep_sh[mt], ep_base);
                        }
                        lds_read_f4_off<768 + mt * 64>(ep_nb[mt], ep_base);
                    });
                static_for<4 * NT>([&](auto j_c) {
                    constexpr int j = decltype(j_c)::value;
                    constexpr int nt = j >> 2, mt = j & 3;
                    constexpr int issued_next = RPS * j < n_next ? RPS * j : n_next; // reads of the next k-step issued so far
                    if constexpr (mt == 0) {
                        // reads of THIS k-step still allowed in flight: those after (the last dword of) B_nt; plus all reads
                        // of the next one issued so far.  (First k-step of a later chunk: its B came before the barrier, its
                        // A after -> everything of this k-step must be in.)
                        constexpr int after = (first_of_chunk && part > 0) ? 0 : (cur_gather ? 4 * (NT - 1 - nt) : NT - 1 - nt);
                        wait_lgkm(after + issued_next + n_ep);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    static_for<RPS>([&](auto rr_c) { // reads of the next k-step, in its read order
                        constexpr int r0 = RPS * j + decltype(rr_c)::value;
                        if constexpr (r0 < n_next) {
                            constexpr int r = more_here ? r0 : r0 + 4; // a B-only prefetch skips the A slots
                            if constexpr (r < 4) READ_A(a[nxt][r], r < 3 ? wbl : wbl3, (ksl + 1) * REC + r * 1024);
                            else if constexpr (next_gather)
                                read_bsp(bsp[(r - 4) / 4][(r - 4) % 4], std::integral_constant<int, (r - 4) % 4>{}, std::integral_constant<int, (r - 4) / 4>{});
                            else read_b(b[nxt][r - 4], std::integral_constant<int, ks_next>{}, std::integral_constant<int, r - 4>{});
                        }
                    });
                    if constexpr (cur_gather) {
                        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                        const u32x4 u = {bsp[nt][0], bsp[nt][1], bsp[nt][2], bsp[nt][3]};
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[cur][mt], __builtin_bit_cast(half8, u), acc[mt][nt], 0, 0, 0);
                    } else {
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[cur][mt], b[cur][nt], acc[mt][nt], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
            chunk++;
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // nothing of mine in flight when the epilogue touches LDS
        static_for<4>([&](auto mt_c) { // the prefetched parameters stay allocated until they have landed
            constexpr int mt = decltype(mt_c)::value;
            if constexpr (!IS_FIRST) {
                keep_alive(ep_sc[mt]);
                keep_alive(ep_sh[mt]);
            }
            keep_alive(ep_nb[mt]);
        });
        // ---- epilogue of this conv (the wave's own boards only: no barrier needed) ------------------------
        // The accumulators were initialised with this conv's bias, so: conv1: u = lrelu(acc); conv2: x += acc,
        // a = lrelu(scale*x + shift).  LeakyReLU runs on the packed fp16 values (v_pk_mul_f16 + v_pk_max_f16).
        // Three straight-line variants picked ONCE per conv (left to the compiler the uniform conditions were
        // re-tested, with branches and exec masking, for every tile); stores are unconditional: padding lanes and
        // the non-existent 8th octet go to a per-lane trash slot.
        auto epilogue = [&](auto kind) {
            constexpr int KIND = decltype(kind)::value; // 0: conv1, 1: conv2 (not last), 2: last conv
#pragma unroll
            for (int mt = 0; mt < 4; mt++) {
                const int co0 = 16 * mt + 4 * q;
                const bool wr = (2 * mt + (q >> 1)) < N_OCT;
                const int woff = (2 * mt + (q >> 1)) * plane_b + (q & 1) * 8; // octet plane + half of the octet
                const f32x4 sc = ep_sc[mt], sh = ep_sh[mt], next_bias = ep_nb[mt];
#pragma unroll
                for (int nt = 0; nt < NT; nt++) {
                    f32x4 v = acc[mt][nt];
                    acc[mt][nt] = next_bias;
                    half4 o;
                    if (KIND == 0) {
                        o = lrelu_h4(__builtin_convertvector(v, half4));
                    } else {
                        f32x4 xv = xres[mt][nt] + v;
                        xres[mt][nt] = xv;
                        if (KIND == 2) {
                            o = __builtin_convertvector(xv, half4);
                            if (grow[nt] >= 0 && co0 < p.xout_c) *(half4 *)(p.xout + (size_t)grow[nt] * p.xout_c + co0) = o;
                            continue;
                        }
                        o = lrelu_h4(__builtin_convertvector(__builtin_elementwise_fma(sc, xv, sh), half4)); // one v_pk_fma_f32 per pair
                    }
                    if (L15 && mt == 3) { // channels 48, 49 (lanes q = 0) -> the compact plane; 50..63 do not exist
                        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                        const int wa = (q == 0 && grow[nt] >= 0) ? p6_addr[nt] : trash;
                        *(unsigned *)(lds + wa) = __builtin_bit_cast(u32x2, o)[0];
                    } else {
                        const int wa = (wr && grow[nt] >= 0) ? pos_addr[nt] + woff : trash;
                        *(half4 *)(lds + wa) = o;
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
