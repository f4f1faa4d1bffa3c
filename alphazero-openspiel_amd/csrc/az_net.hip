// az_net.hip — fused PV-net inference for gfx950 (MI355X): the whole residual tower in ONE kernel.
//
// Reference computation: Net.forward / ResidualBlock.forward (network.py:48-64,99-104) in eval mode.
// Mapping (weights pre-packed by alphazero-openspiel_amd/fusednet.py, see include/az_net.h):
//   * a wavefront owns BPW whole boards; their activations never leave the CU: one fp16 LDS image
//     [board][cell][56 ch] with a zero halo (cell = (y+1)*(W+1) + (x+1); the halo column is shared between
//     rows), rewritten in place layer after layer; the fp32 residual stream lives in registers.
//   * every 3x3 conv is an implicit GEMM on v_mfma_f32_16x16x32_f16:  D[co][n] += Wp[co][k] * Act[k][n],
//     n = (board, position) over the wave's boards, k = 64 groups x 8 channels (group -> tap, channel
//     octet; group 63 = zero padding); conv 0, which sees only the input planes, is compacted on upload to
//     16 groups (9 taps x one octet) = 4 k-steps.  M = 64 output channels (4 tiles),
//     so the accumulator of lane l holds 4 CONSECUTIVE channels of one position: the epilogue
//     (bias, LeakyReLU, next BN scale/shift) packs them to fp16 and writes 8 bytes back to the image.
//   * weights are the A operand, shared by all waves of the workgroup: streamed L2 -> LDS by
//     global_load_lds (16 B/lane) in 16 KiB chunks (4 k-steps), double buffered, one barrier per chunk.
//     They are stored fragment-linear, so an A fragment is one contiguous KiB (conflict-free ds_read_b128).
// The fc1 + softmax + tanh head is a second small MFMA kernel over the tower output (fp16, [B][HW][64]).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../include/az_engine.h"
#include "../../include/az_net.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Timing-only ablation switches (make ABL="-DAZ_ABL_..."): each removes one ingredient of the tower's inner
// structure so its cost can be read off the clock.  Outputs are wrong by construction; never shipped.
//   AZ_ABL_NOEPI   no epilogue arithmetic / LDS writes        AZ_ABL_NOB    no B-fragment (activation) LDS reads
//   AZ_ABL_NOA     no A-fragment (weight) LDS reads           AZ_ABL_NODMA  no weight DMA and no chunk barrier
//   AZ_ABL_NOBARRIER  chunk barriers dropped (the DMA stays)   AZ_ABL_SKEW=n waves 4..7 start n x 64 cycles late
//   (round 2: NOBARRIER alone and with SKEW = 30 / 60 - the two waves of a SIMD running a third / half a conv apart -
//    all time within noise of the shipped kernel: de-phasing the wave pairs buys nothing)
#define OCT_B 16 // one cell of one channel-octet plane: 8 fp16
#define AZ_NET_K0STEPS 4 // k-steps of conv 0 on the device (9 taps x the one octet holding the input planes, padded to 16 groups)
#define N_OCT 7  // 56 channels
#define AZ_MAX_DEVICES 64

struct TowerParams {
    int H, W, HW, cells, cin, n_convs, n_boards, bpw;
    int rcells;  // cells per wave region (bpw boards + zero pad), multiple of 16
    int zcell;   // a cell whose whole 3x3 neighbourhood is never written (reads of padding columns land here)
    int rs;      // row stride of the cell grid: 8 when W <= 7 ("row-pair tiles"), else W + 1
    int tpb;     // row-pair mode: column tiles per board = ceil(H / 2); 0 = generic column packing
    int off_epi; // LDS byte offset of the epilogue-parameter ring: 2 slots x [4][64] floats (scale, shift, next bias)
    int off_act; // LDS byte offset of the activation planes
    const _Float16 *conv_w;
    const float *epi;    // [n_convs][4][64]: bias, next-prologue scale, shift, bias of the NEXT conv
    const float *skip_w; // [64][4]
    float in_scale[8], in_shift[8];
    const float *obs;
    _Float16 *xout;
    _Float16 *xout_lo; // f16x3: lo halves of the tower output
};

__device__ __forceinline__ float lrelu(float v) { return fmaxf(v, 0.01f * v); }

__device__ __forceinline__ half4 lrelu_h4(half4 h) { return __builtin_elementwise_max(h, h * (_Float16)0.01f); }
__device__ __forceinline__ int opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}
// counted LDS wait with a literal operand (n folds to a constant after unrolling)
__device__ __forceinline__ void wait_lgkm(int n) {
    switch (n < 15 ? n : 15) {
    case 0: asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt lgkmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt lgkmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt lgkmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt lgkmcnt(13)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt lgkmcnt(14)" ::: "memory"); break;
    default: asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory"); break;
    }
}
// LDS fragment read the compiler does not track (no automatic s_waitcnt): waited for by hand in the k-loop
__device__ __forceinline__ void lds_read128(half8 &dst, unsigned lds_byte_addr) {
    asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(lds_byte_addr));
}
// same, with a compile-time byte offset in the instruction's 16-bit offset field (no address arithmetic in the loop)
template <int OFF> __device__ __forceinline__ void lds_read128_off(half8 &dst, unsigned lds_byte_addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_byte_addr), "n"(OFF));
}
template <int OFF> __device__ __forceinline__ void lds_read_f4_off(f32x4 &dst, unsigned lds_byte_addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_byte_addr), "n"(OFF));
}
__device__ __forceinline__ void keep_alive(const f32x4 &v) { asm volatile("" ::"v"(v)); }
// (ablation stand-in for a fragment read: defines the register, touches nothing)
__device__ __forceinline__ void fake_read128(half8 &dst, unsigned lds_byte_addr) { asm volatile("" : "=v"(dst) : "v"(lds_byte_addr)); }
#ifdef AZ_ABL_NOA
#define READ_A(dst, addr, off) fake_read128(dst, addr)
#else
#define READ_A(dst, addr, off) lds_read128_off<(off)>(dst, addr)
#endif
template <int OFF> __device__ __forceinline__ void lds_read32_off(unsigned &dst, unsigned lds_byte_addr) {
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_byte_addr), "n"(OFF));
}
__device__ __forceinline__ void fake_read32(unsigned &dst, unsigned lds_byte_addr) { asm volatile("" : "=v"(dst) : "v"(lds_byte_addr)); }
#ifdef AZ_ABL_NOB
#define READ_B32_OFF(dst, addr, off) fake_read32(dst, addr)
#else
#define READ_B32_OFF(dst, addr, off) lds_read32_off<(off)>(dst, addr)
#endif
#ifdef AZ_ABL_NOB
#define READ_B(dst, addr) fake_read128(dst, addr)
#define READ_B_OFF(dst, addr, off) fake_read128(dst, addr)
#else
#define READ_B(dst, addr) lds_read128(dst, addr)
#define READ_B_OFF(dst, addr, off) lds_read128_off<(off)>(dst, addr)
#endif
// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{})
template <class F, int... I> __device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

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
    // ---- per-lane tables: the NT*16 columns of this wave -----------------------------------------------------
    int pos_addr[NT], grow[NT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        int b, y, x;
        bool ok;
        if (p.tpb) { // row-pair tile
            b = nt / p.tpb;
            y = 2 * (nt - b * p.tpb) + (l15 >> 3);
            x = l15 & 7;
            ok = x < p.W && y < p.H && b < p.bpw; // (a kernel with more tiles than the boards need masks the rest)
        } else {
            int n = nt * 16 + l15;
            b = n / p.HW;
            int pos = n - b * p.HW;
            y = pos / p.W;
            x = pos - y * p.W;
            ok = b < p.bpw;
        }
        ok = ok && (board0 + b < p.n_boards);
        int cell = b * p.cells + (y + 1) * p.rs + (x + 1);
        pos_addr[nt] = region + ((ok || p.tpb) ? cell : p.zcell) * OCT_B; // row-pair padding lanes read their (finite) neighbours
        grow[nt] = ok ? (board0 + b) * p.HW + y * p.W + x : -1;
    }
    int koff[AZ_NET_KSTEPS]; // byte offset (tap shift + octet plane) of this lane's k-group in each k-step
#pragma unroll
    for (int ks = 0; ks < AZ_NET_KSTEPS; ks++) {
        int g = 4 * ks + q, tap, c8;
        bool zero;
        if (L15) {
            tap = g / 6, c8 = g - tap * 6;
            zero = g >= 54;
        } else {
            tap = g / 7, c8 = g - tap * 7;
            zero = g == 63;
        }
        int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        koff[ks] = zero ? 0 : (dy * p.rs + dx) * OCT_B + c8 * plane_b; // zero groups: zero weights, any finite data
        if (RP1) koff[ks] += (int)lds_base + pos_addr[0];              // the full LDS address of tile 0's fragment
    }
    // L15: channels 48, 49 live in a COMPACT plane - 4 bytes per cell in the space of octet plane 6 - so that the 4-byte
    // gather reads of the last k-step touch 16 consecutive dwords per 16 columns (at the octet planes' 16-byte cell stride
    // they were 4-way bank conflicted: 17 % of the kernel's LDS cycles, profiles/r2_bench_default_pmc_summary.txt)
    int p6_addr[NT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) p6_addr[nt] = region + 6 * plane_b + ((pos_addr[nt] - region) >> 2);
    int ksp[4]; // the four taps of this lane's group in the last k-step
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int tap = 4 * q + i;
        tap = tap > 8 ? 8 : tap; // (taps past the ninth carry zero weights)
        int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        ksp[i] = (dy * p.rs + dx) * 4;
        if (RP1) ksp[i] += (int)lds_base + p6_addr[0];
    }
    // conv 0 reads the 4 input planes only (octet 0): K = 9 taps x 1 octet, packed as ONE 4-k-step chunk
    // (group g < 9 = tap g of octet 0, groups 9..15 zero weights) instead of 16 k-steps that are 6/7 zeros.
    int koff0[AZ_NET_K0STEPS];
#pragma unroll
    for (int ks = 0; ks < AZ_NET_K0STEPS; ks++) {
        int g = 4 * ks + q;
        int dy = g / 3 - 1, dx = g - (g / 3) * 3 - 1;
        koff0[ks] = g < 9 ? (dy * p.rs + dx) * OCT_B : 0;
        if (RP1) koff0[ks] += (int)lds_base + pos_addr[0];
    }

    f32x4 acc[4][NT], xres[4][NT];
#ifdef AZ_ABL_SKEW // (timing-only) waves 4..7 start AZ_ABL_SKEW x 64 cycles late
    if (wave >= 4)
        for (int i = 0; i < AZ_ABL_SKEW; i++) __builtin_amdgcn_s_sleep(1);
#endif
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
#ifndef AZ_ABL_NODMA
    if (wave == 0) // conv 0 has a single chunk: its epilogue parameters must land before that chunk's barrier
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const unsigned char *)p.epi + lane * 16),
                                         (__attribute__((address_space(3))) void *)(lds + p.off_epi), 16, 0, 0);
#endif

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
#ifndef AZ_ABL_NODMA
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifndef AZ_ABL_NOBARRIER // (timing-only experiment: how much would de-phasing the two waves of a SIMD be worth?)
            __syncthreads();
#endif
            if (chunk + 1 < n_chunks) issue_chunk(chunk + 1);
            if (!IS_FIRST && part == 0 && wave == 0) // this conv's epilogue parameters ride the same DMA path into a 2-slot ring;
                                                     // they land before the next chunk barrier, long before the epilogue reads them
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const unsigned char *)p.epi + (size_t)conv * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void *)(lds + p.off_epi + (conv & 1) * 1024), 16, 0, 0);
#endif
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
#ifdef AZ_ABL_NOEPI
                    asm volatile("" ::"v"(v));
                    if (KIND == 2 && grow[nt] >= 0) *(half4 *)(p.xout + (size_t)grow[nt] * AZ_NET_XOUT_C + co0) = __builtin_convertvector(v, half4);
                    continue;
#endif
                    half4 o;
                    if (KIND == 0) {
                        o = lrelu_h4(__builtin_convertvector(v, half4));
                    } else {
                        f32x4 xv = xres[mt][nt] + v;
                        xres[mt][nt] = xv;
                        if (KIND == 2) {
                            o = __builtin_convertvector(xv, half4);
                            if (grow[nt] >= 0) *(half4 *)(p.xout + (size_t)grow[nt] * AZ_NET_XOUT_C + co0) = o;
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
#define X3_LOFF_RP1 (N_OCT * 96 * OCT_B) // row-pair boards (W <= 7, H <= 6): lo planes sit a compile-time distance after hi
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
    int pos_addr[NT], grow[NT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        int y, x;
        bool ok;
        if (p.tpb) {
            y = 2 * nt + (l15 >> 3);
            x = l15 & 7;
            ok = x < p.W && y < p.H && nt < p.tpb;
        } else {
            int pos = nt * 16 + l15;
            y = pos / p.W;
            x = pos - y * p.W;
            ok = pos < p.HW;
        }
        ok = ok && board0 < p.n_boards;
        int cell = (y + 1) * p.rs + (x + 1);
        pos_addr[nt] = region + ((ok || p.tpb) ? cell : p.zcell) * OCT_B;
        grow[nt] = ok ? board0 * p.HW + y * p.W + x : -1;
    }
    int koff[AZ_NET_KSTEPS];
#pragma unroll
    for (int ks = 0; ks < AZ_NET_KSTEPS; ks++) {
        int g = 4 * ks + q, tap, c8;
        bool zero;
        if (L15) {
            tap = g / 6, c8 = g - tap * 6;
            zero = g >= 54;
        } else {
            tap = g / 7, c8 = g - tap * 7;
            zero = g == 63;
        }
        int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        koff[ks] = zero ? 0 : (dy * p.rs + dx) * OCT_B + c8 * plane_b;
        if (RP1) koff[ks] += (int)lds_base + pos_addr[0];
    }
    int p6_addr[NT]; // compact plane of channels 48, 49 (4 bytes per cell, see az_tower_kernel); its lo twin at + lo_off
#pragma unroll
    for (int nt = 0; nt < NT; nt++) p6_addr[nt] = region + 6 * plane_b + ((pos_addr[nt] - region) >> 2);
    int ksp[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int tap = 4 * q + i;
        tap = tap > 8 ? 8 : tap;
        int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        ksp[i] = (dy * p.rs + dx) * 4;
        if (RP1) ksp[i] += (int)lds_base + p6_addr[0];
    }
    int koff0[AZ_NET_K0STEPS];
#pragma unroll
    for (int ks = 0; ks < AZ_NET_K0STEPS; ks++) {
        int g = 4 * ks + q;
        int dy = g / 3 - 1, dx = g - (g / 3) * 3 - 1;
        koff0[ks] = g < 9 ? (dy * p.rs + dx) * OCT_B : 0;
        if (RP1) koff0[ks] += (int)lds_base + pos_addr[0];
    }

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
                    acc[mt][nt] = next_bias;
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

// ------------------------------------------------------------------------------------------------
// fc1 + softmax + tanh (network.py:61-64).  One workgroup = 16 boards; the K = HW*64 reduction is split
// over the 4 waves (k-step ks goes to wave ks & 3), partial tiles are summed through LDS.
struct HeadParams {
    int HW, A, n_ot, ksteps, n_boards;
    const _Float16 *x;    // [B][HW*64]
    const _Float16 *fc_w; // [n_ot][ksteps][64][8]
    const _Float16 *x_lo, *fc_w_lo; // f16x3: the lo halves (scaled by 2048), same layouts
    const float *fc_b;
    float *priors, *values;
};

#define OTG 8
#define HEAD_NW 8 // waves per workgroup: the K reduction is split over them (memory-bound: more loads in flight per CU)
// X3: split-fp16 operands (see az_tower_x3_kernel): three MFMAs per product, result = acc + acc2 / 2048.
template <bool X3> __global__ __launch_bounds__(HEAD_NW * 64) void az_head_kernel(HeadParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float *part = (float *)lds;                               // [HEAD_NW waves][OTG][64 lanes][4]
    float *logits = (float *)(lds + HEAD_NW * OTG * 64 * 16); // [16][n_ot*16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, l15 = lane & 15;
    const int b0 = blockIdx.x * 16;
    const int K = p.HW * AZ_NET_XOUT_C, NP = p.n_ot * 16;
    int row = b0 + l15;
    if (row >= p.n_boards) row = p.n_boards - 1; // clamp: computed, never stored
    const _Float16 *xrow = p.x + (size_t)row * K + 8 * q;
    const _Float16 *xrow_lo = X3 ? p.x_lo + (size_t)row * K + 8 * q : nullptr;
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
                half8 a = *(const half8 *)(xrow + 32 * ks);
                half8 w = *(const half8 *)(p.fc_w + ((size_t)ks * 64 + lane) * 8);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, w, acc[0], 0, 0, 0);
                if constexpr (X3) {
                    half8 al = *(const half8 *)(xrow_lo + 32 * ks);
                    half8 wl = *(const half8 *)(p.fc_w_lo + ((size_t)ks * 64 + lane) * 8);
                    acc2[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, wl, acc2[0], 0, 0, 0);
                    acc2[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, w, acc2[0], 0, 0, 0);
                }
            }
        } else
        for (int ks = wave; ks < p.ksteps; ks += HEAD_NW) {
            half8 a = *(const half8 *)(xrow + 32 * ks);
            half8 al;
            if constexpr (X3) al = *(const half8 *)(xrow_lo + 32 * ks);
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
#define HEAD_OTG 4  // output tiles (x16 outputs) per workgroup
#define HEAD_RING 4 // weight chunks resident in LDS: chunk c is multiplied while c+1 .. c+RING-2 are in flight
// HEAD_MT: board tiles (x16 boards) per wave - every weight fragment read from LDS feeds HEAD_MT MFMAs
template <bool X3, int HEAD_MT> __global__ __launch_bounds__(256) void az_head_logits_kernel(HeadParams p, float *__restrict__ logits_g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int CK = X3 ? 2 : 4;                       // k-steps per chunk: 16 KiB of weight fragments either way
    constexpr int NPART = X3 ? 2 : 1;
    constexpr int FRAGS = CK * HEAD_OTG * NPART;         // KiB fragments per chunk: [part][ksl][o]
    constexpr int CHUNK_B = FRAGS * 1024;
    constexpr int PER = FRAGS / 4 + CK * NPART * HEAD_MT; // vector-memory operations one wave issues per chunk
    static_assert(FRAGS % 4 == 0 && 2 * PER < 64, "pieces split evenly over the 4 waves; the counted waits fit vmcnt");
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
        int row = b0 + 16 * m + l15;
        if (row >= p.n_boards) row = p.n_boards - 1; // clamp: computed, never stored
        xrow[m] = p.x + (size_t)row * K + 8 * q;
        xrow_lo[m] = X3 ? p.x_lo + (size_t)row * K + 8 * q : nullptr;
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
                if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
                else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                // a BARE barrier: __syncthreads() carries a fence that drains vmcnt to 0 and with it the chunks in flight
                __builtin_amdgcn_s_barrier(); // everybody's pieces of chunk c are in LDS, and the slot of chunk c - 1 is free again
                asm volatile("" ::: "memory");
                if (c + HEAD_RING - 1 < n_chunks) issue_chunk(c + HEAD_RING - 1, std::integral_constant<int, (slot + HEAD_RING - 1) % HEAD_RING>{});
                const unsigned char *wb = lds + slot * CHUNK_B + lane * 16;
#pragma unroll
                for (int ksl = 0; ksl < CK; ksl++) {
                    if (c * CK + ksl < p.ksteps) {
#pragma unroll
                        for (int o = 0; o < HEAD_OTG; o++) {
                            const half8 w = *(const half8 *)(wb + (ksl * HEAD_OTG + o) * 1024);
                            half8 wl;
                            if constexpr (X3) wl = *(const half8 *)(wb + ((CK + ksl) * HEAD_OTG + o) * 1024);
#pragma unroll
                            for (int m = 0; m < HEAD_MT; m++) {
                                acc[m][o] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[slot][ksl][m], w, acc[m][o], 0, 0, 0);
                                if constexpr (X3) {
                                    acc2[m][o] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[slot][ksl][m], wl, acc2[m][o], 0, 0, 0);
                                    acc2[m][o] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[slot][ksl][m], w, acc2[m][o], 0, 0, 0);
                                }
                            }
                        }
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
        const float bias = p.fc_b[col];
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

// ================================================================================================
struct az_net {
    az_net_desc d;
    std::string err;
    _Float16 *conv_w = nullptr, *fc_w = nullptr, *xout = nullptr;
    _Float16 *fc_w_lo = nullptr, *xout_lo = nullptr; // f16x3 only
    float *epi = nullptr, *fc_b = nullptr, *skip_w = nullptr, *logits = nullptr;
    float in_affine[16];
    int max_boards = 0;
    int bpw_max = 0, lds_head = 0, n_ot = 0, r3 = 16;
    int precision = AZ_NET_PREC_F16;
};
static std::string g_net_err;

#define NCHK(n, call)                                                        \
    do {                                                                     \
        hipError_t _s = (call);                                              \
        if (_s != hipSuccess) {                                              \
            (n)->err = std::string(#call) + ": " + hipGetErrorString(_s);    \
            return AZ_E_HIP;                                                 \
        }                                                                    \
    } while (0)

extern "C" const char *az_net_last_error(const az_net *n) { return n ? n->err.c_str() : g_net_err.c_str(); }

extern "C" int az_net_destroy(az_net *n) {
    if (!n) return AZ_OK;
    (void)hipSetDevice(n->d.device);
    (void)hipFree(n->conv_w);
    (void)hipFree(n->fc_w);
    (void)hipFree(n->fc_w_lo);
    (void)hipFree(n->xout_lo);
    (void)hipFree(n->xout);
    (void)hipFree(n->epi);
    (void)hipFree(n->fc_b);
    (void)hipFree(n->skip_w);
    (void)hipFree(n->logits);
    delete n;
    return AZ_OK;
}

// geometry of one launch for a given boards-per-wave
struct TowerGeom {
    int bpw, nt, ck, waves, rcells, zcell, rs, tpb, cells, off_epi, off_act, lds;
};
static TowerGeom tower_geom(int bpw, int waves, int H, int W) {
    TowerGeom g;
    g.bpw = bpw;
    if (W <= 7) { // row-pair tiles at row stride 8 (conflict-free B reads)
        g.rs = 8;
        g.tpb = (H + 1) / 2;
        g.nt = bpw * g.tpb;
    } else {
        g.rs = W + 1;
        g.tpb = 0;
        g.nt = (bpw * H * W + 15) / 16;
    }
    g.cells = (H + 2) * g.rs + 1;
    int zpad = 2 * (g.rs + 1) + 1;
    g.rcells = (bpw * g.cells + zpad + 15) & ~15;
    g.zcell = bpw * g.cells + (g.rs + 1);
    int wave_act = N_OCT * g.rcells * OCT_B;
    // waves = 8 (512 threads): small tiles only (<= 256 registers/wave), 2 waves per SIMD, so one wave's epilogue / waits
    // overlap another's MFMAs, and the eight share ONE weight stream (two co-resident 4-wave workgroups each pull their
    // own: at 1 board per wave that is ~25 TB/s of L2 reads chip-wide).
    g.waves = waves;
    int act = g.waves * wave_act;
    // 32 KiB weight chunks (half the barriers) when LDS allows; not with 8 waves: the longer unrolled body spills there
    g.ck = (act + 6144 + 2 * 8 * 4096 <= 160 * 1024) ? 8 : 4; // 32 KiB chunks (half the barriers) when LDS allows

    g.off_epi = 2 * g.ck * 4096;
    g.off_act = g.off_epi + 2048 + 8 * 64 * 8; // epilogue ring (2 KiB) + trash slots (8 B per thread, up to 512 threads)
    g.lds = g.off_act + act;
    return g;
}

// geometry of the f16x3 tower: one board per wave, 4 waves, 4-k-step chunks of (hi, lo) records
struct X3Geom {
    int nt, rcells, zcell, rs, tpb, cells, off_epi, off_act, lds, lo_off;
    bool rp1;
};
static X3Geom x3_geom(int H, int W, int r3) {
    X3Geom g;
    if (W <= 7) {
        g.rs = 8;
        g.tpb = (H + 1) / 2;
        g.nt = g.tpb;
    } else {
        g.rs = W + 1;
        g.tpb = 0;
        g.nt = (H * W + 15) / 16;
    }
    g.cells = (H + 2) * g.rs + 1;
    int zpad = 2 * (g.rs + 1) + 1;
    g.rcells = (g.cells + zpad + 15) & ~15;
    g.zcell = g.cells + (g.rs + 1);
    g.rp1 = g.tpb && g.rs == 8 && g.tpb <= 3 && g.rcells <= 96;
    const int region_b = N_OCT * g.rcells * OCT_B;
    g.lo_off = g.rp1 ? X3_LOFF_RP1 : region_b;
    const int rows = r3 < 16 ? r3 + 1 : 16, rec = 3 * 1024 + 4 * rows * 16;
    const int chunk_s = (4 * 2 * rec + 1023) & ~1023;
    g.off_epi = 2 * chunk_s;
    g.off_act = g.off_epi + 2048 + 256 * 16;
    g.lds = g.off_act + 4 * 2 * g.lo_off;
    if (g.nt < 3) g.nt = 3;
    return g;
}

extern "C" int az_net_create(const az_net_desc *desc, az_net **out) {
    if (!desc || !out) {
        g_net_err = "null argument";
        return AZ_E_INVALID;
    }
    *out = nullptr;
    if (desc->struct_size != (int32_t)sizeof(az_net_desc)) {
        g_net_err = "az_net_desc.struct_size mismatch";
        return AZ_E_INVALID;
    }
    const az_net_desc &d = *desc;
    if (d.rows < 3 || d.cols < 3 || d.rows * d.cols > 64 || d.in_planes < 1 || d.in_planes > 4 || d.n_filters < 1 ||
        d.n_filters > AZ_NET_CPAD || d.n_blocks < 1 || d.num_actions < 1 || !d.conv_w || !d.conv_epi || !d.in_affine ||
        !d.skip_w || !d.fc_w || !d.fc_b) {
        g_net_err = "bad net description (need 3<=rows,cols, rows*cols<=64, in_planes<=4, n_filters<=56, packed buffers)";
        return AZ_E_INVALID;
    }
    if (d.precision != AZ_NET_PREC_F16 && d.precision != AZ_NET_PREC_F16X3) {
        g_net_err = "precision must be AZ_NET_PREC_F16 or AZ_NET_PREC_F16X3";
        return AZ_E_INVALID;
    }
    if (d.precision == AZ_NET_PREC_F16X3 && (!d.conv_w_lo || !d.fc_w_lo)) {
        g_net_err = "AZ_NET_PREC_F16X3 needs conv_w_lo and fc_w_lo";
        return AZ_E_INVALID;
    }
    az_net *n = new az_net();
    n->d = d;
    n->precision = d.precision;
    memcpy(n->in_affine, d.in_affine, sizeof n->in_affine);
    const int HW = d.rows * d.cols;
    if (n->precision == AZ_NET_PREC_F16X3) {
        X3Geom g = x3_geom(d.rows, d.cols, d.n_filters <= 50 ? 2 : 16);
        if (g.nt > 4 || g.lds > 160 * 1024) {
            g_net_err = "board / filter count does not fit the f16x3 tower kernel's LDS budget";
            delete n;
            return AZ_E_INVALID;
        }
    }
    // boards per wave: at most 4 column tiles per wave (larger tiles spill registers under the hand-scheduled k-loop
    // and measured slower than more, smaller waves) within the 160 KiB LDS
    int best = 0;
    for (int bpw = 1; bpw <= 8; bpw++) {
        TowerGeom g = tower_geom(bpw, 4, d.rows, d.cols);
        if (g.nt > 4 || g.lds > 160 * 1024) break;
        best = bpw;
    }
    if (!best) {
        g_net_err = "board does not fit the tower kernel's LDS budget";
        delete n;
        return AZ_E_INVALID;
    }
    n->bpw_max = best;
    n->n_ot = (d.num_actions + 1 + 15) / 16;
    n->lds_head = HEAD_NW * OTG * 64 * 16 + 16 * n->n_ot * 16 * 4;
    hipError_t s = hipSetDevice(d.device);
    if (s != hipSuccess) {
        g_net_err = std::string("hipSetDevice: ") + hipGetErrorString(s);
        delete n;
        return AZ_E_HIP;
    }
    size_t fw = (size_t)n->n_ot * (HW * AZ_NET_XOUT_C / 32) * 64 * 8 * 2, fb = (size_t)n->n_ot * 16 * 4;
    int rc = AZ_OK;
    auto up = [&](void **dst, const void *src, size_t bytes) {
        if (rc != AZ_OK) return;
        if (hipMalloc(dst, bytes) != hipSuccess || hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) != hipSuccess) {
            g_net_err = "hipMalloc/hipMemcpy of the packed weights failed";
            rc = AZ_E_NOMEM;
        }
    };
    {   // device layout: [conv 0 compacted to AZ_NET_K0STEPS k-steps][conv 1 ..][..], one RECORD per k-step: output-channel
        // tiles 0..2 as in the ABI layout (3 KiB), tile 3 with only its stored rows (see WRec).  Conv 0 sees the input
        // planes only (channel octet 0), i.e. ABI groups 7*tap; they become groups 0..8 of its 4 k-steps.
        // <= 50 filters: the K dimension is re-grouped into 15 k-steps (see dev_group below) - channels 48, 49 of the nine
        // taps fit ONE k-step instead of filling the seventh channel octet of every tap.
        n->r3 = d.n_filters <= 50 ? 2 : 16;
        const bool l15 = n->r3 < 16;
        const int nks = l15 ? 15 : AZ_NET_KSTEPS;
        const int rows = n->r3 < 16 ? n->r3 + 1 : 16, rec = 3 * 1024 + 4 * rows * 16;
        const int n_convs = 2 * d.n_blocks;
        const size_t conv_b = (size_t)AZ_NET_KSTEPS * 4096;
        const size_t n_rec = (size_t)AZ_NET_K0STEPS + (size_t)(n_convs - 1) * nks;
        auto build_records = [&](const unsigned char *src, std::vector<unsigned char> &dev) { // one `rec`-byte record per k-step
        dev.assign(n_rec * rec, 0);
        auto put_record = [&](unsigned char *dst, const unsigned char *ks4k) { // ks4k: [4 mt][64 lanes][16 B]
            memcpy(dst, ks4k, 3 * 1024);
            for (int q = 0; q < 4; q++)
                for (int r = 0; r < (n->r3 < 16 ? n->r3 : 16); r++)
                    memcpy(dst + 3 * 1024 + (q * rows + r) * 16, ks4k + 3 * 1024 + (q * 16 + r) * 16, 16);
        };
        // ABI: element j of group g = tap * 7 + c8 is channel 8 * c8 + j at that tap.  abi_half: one fp16 of a conv
        auto abi_half = [&](int c, int mt, int l15_, int tap, int ch) -> uint16_t {
            int g = tap * 7 + (ch >> 3), oks = g >> 2, olane = (g & 3) * 16 + l15_;
            const uint16_t *w = (const uint16_t *)(src + (size_t)c * conv_b);
            return w[((((size_t)oks * 4 + mt) * 64 + olane) * 8) + (ch & 7)];
        };
        std::vector<unsigned char> c0(AZ_NET_K0STEPS * 4096, 0); // conv 0 compacted, still in 4 KiB k-steps
        for (int ks = 0; ks < AZ_NET_K0STEPS; ks++)
            for (int mt = 0; mt < 4; mt++)
                for (int lane = 0; lane < 64; lane++) {
                    int g = 4 * ks + (lane >> 4);
                    if (g >= 9) continue;
                    int go = 7 * g, oks = go >> 2, olane = (go & 3) * 16 + (lane & 15);
                    memcpy(&c0[(((size_t)ks * 4 + mt) * 64 + lane) * 16], src + (((size_t)oks * 4 + mt) * 64 + olane) * 16, 16);
                }
        size_t off = 0;
        for (int ks = 0; ks < AZ_NET_K0STEPS; ks++, off += rec) put_record(&dev[off], &c0[(size_t)ks * 4096]);
        std::vector<uint16_t> k4(4096 / 2);
        for (int c = 1; c < n_convs; c++)
            for (int ks = 0; ks < nks; ks++, off += rec) {
                if (!l15) {
                    put_record(&dev[off], src + (size_t)c * conv_b + (size_t)ks * 4096);
                    continue;
                }
                // 15-k-step grouping: group g' = 4 ks + q.  g' < 54: (tap, octet) = divmod(g', 6), the 48 channels of six
                // full octets; g' = 54, 55: zero; k-step 14: element j of group q < 3 is channel 48 + (j & 1) at tap
                // 4 q + j / 2 (taps > 8: zero), group 3 zero.
                std::fill(k4.begin(), k4.end(), (uint16_t)0);
                for (int mt = 0; mt < 4; mt++)
                    for (int lane = 0; lane < 64; lane++) {
                        const int q = lane >> 4, l = lane & 15, gp = 4 * ks + q;
                        uint16_t *o = &k4[(((size_t)mt * 64) + lane) * 8];
                        for (int j = 0; j < 8; j++) {
                            if (ks < 14) {
                                if (gp < 54) o[j] = abi_half(c, mt, l, gp / 6, 8 * (gp % 6) + j);
                            } else if (q < 3) {
                                int tap = 4 * q + (j >> 1);
                                if (tap < 9) o[j] = abi_half(c, mt, l, tap, 48 + (j & 1));
                            }
                        }
                    }
                put_record(&dev[off], (const unsigned char *)k4.data());
            }
        };
        std::vector<unsigned char> hi, dev;
        build_records((const unsigned char *)d.conv_w, hi);
        const size_t pad = 2 * 8 * 4096 + 1024; // a chunk of padding: the last (short) chunk is fetched at full length
        if (n->precision == AZ_NET_PREC_F16X3) { // per k-step: hi record, lo record
            std::vector<unsigned char> lo;
            build_records((const unsigned char *)d.conv_w_lo, lo);
            dev.assign(2 * n_rec * rec + pad, 0);
            for (size_t r = 0; r < n_rec; r++) {
                memcpy(&dev[2 * r * rec], &hi[r * rec], rec);
                memcpy(&dev[(2 * r + 1) * rec], &lo[r * rec], rec);
            }
        } else {
            dev = hi;
            dev.resize(n_rec * rec + pad, 0);
        }
        up((void **)&n->conv_w, dev.data(), dev.size());
    }
    {   // [conv][3][64] (ABI) -> [conv][4][64] with the NEXT conv's bias in row 3 (what the kernel's ring slot holds)
        int nc = 2 * d.n_blocks;
        std::vector<float> e4((size_t)nc * 256, 0.f);
        for (int c = 0; c < nc; c++) {
            memcpy(&e4[(size_t)c * 256], d.conv_epi + (size_t)c * 192, 192 * sizeof(float));
            if (c + 1 < nc) memcpy(&e4[(size_t)c * 256 + 192], d.conv_epi + (size_t)(c + 1) * 192, 64 * sizeof(float));
        }
        up((void **)&n->epi, e4.data(), e4.size() * sizeof(float));
    }
    up((void **)&n->fc_w, d.fc_w, fw);
    if (n->precision == AZ_NET_PREC_F16X3) up((void **)&n->fc_w_lo, d.fc_w_lo, fw);
    up((void **)&n->fc_b, d.fc_b, fb);
    up((void **)&n->skip_w, d.skip_w, 64 * 4 * sizeof(float));
    if (rc != AZ_OK) {
        az_net_destroy(n);
        return rc;
    }
    n->d.conv_w_lo = nullptr;
    n->d.fc_w_lo = nullptr;
    n->d.conv_w = nullptr; // host pointers are not kept
    n->d.conv_epi = nullptr;
    n->d.in_affine = nullptr;
    n->d.skip_w = nullptr;
    n->d.fc_w = nullptr;
    n->d.fc_b = nullptr;
    *out = n;
    return AZ_OK;
}

extern "C" int az_net_reserve(az_net *n, int32_t max_boards) {
    if (!n || max_boards < 1) return AZ_E_INVALID;
    NCHK(n, hipSetDevice(n->d.device));
    if (n->xout) (void)hipFree(n->xout);
    n->xout = nullptr;
    size_t bytes = (size_t)max_boards * n->d.rows * n->d.cols * AZ_NET_XOUT_C * 2;
    NCHK(n, hipMalloc((void **)&n->xout, bytes));
    NCHK(n, hipMemset(n->xout, 0, bytes));
    if (n->xout_lo) (void)hipFree(n->xout_lo);
    n->xout_lo = nullptr;
    if (n->precision == AZ_NET_PREC_F16X3) {
        NCHK(n, hipMalloc((void **)&n->xout_lo, bytes));
        NCHK(n, hipMemset(n->xout_lo, 0, bytes));
    }
    if (n->logits) (void)hipFree(n->logits);
    n->logits = nullptr;
    if (n->n_ot > OTG) NCHK(n, hipMalloc((void **)&n->logits, (size_t)max_boards * n->n_ot * 16 * sizeof(float)));
    n->max_boards = max_boards;
    return AZ_OK;
}

template <int NT, int CK, int WAVES, bool RP1, int R3> static hipError_t launch_tower_r3(const az_net *n, const TowerParams &tp, int grid, int lds, hipStream_t st) {
    static bool attr_set[AZ_MAX_DEVICES] = {false}; // the attribute is per (function, device)
    const int dv = n->d.device;
    if (dv < 0 || dv >= AZ_MAX_DEVICES || !attr_set[dv]) {
        hipError_t s = hipFuncSetAttribute((const void *)az_tower_kernel<NT, CK, WAVES, RP1, R3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (s != hipSuccess) return s;
        if (dv >= 0 && dv < AZ_MAX_DEVICES) attr_set[dv] = true;
    }
    hipLaunchKernelGGL((az_tower_kernel<NT, CK, WAVES, RP1, R3>), dim3(grid), dim3(WAVES * 64), lds, st, tp);
    return hipGetLastError();
}
template <int NT, int CK, int WAVES, bool RP1> static hipError_t launch_tower_rp(const az_net *n, const TowerParams &tp, int grid, int lds, hipStream_t st) {
    return n->r3 == 2 ? launch_tower_r3<NT, CK, WAVES, RP1, 2>(n, tp, grid, lds, st) : launch_tower_r3<NT, CK, WAVES, RP1, 16>(n, tp, grid, lds, st);
}
template <int NT, int CK, int WAVES> static hipError_t launch_tower(const az_net *n, const TowerParams &tp, int grid, int lds, hipStream_t st) {
    // row-pair tiles (row stride 8), one board per wave, every tile of the wave on that board
    if (tp.tpb && tp.bpw == 1 && tp.rs == 8 && tp.tpb <= NT) return launch_tower_rp<NT, CK, WAVES, true>(n, tp, grid, lds, st);
    return launch_tower_rp<NT, CK, WAVES, false>(n, tp, grid, lds, st);
}
template <int NT> static hipError_t launch_tower_ck(const az_net *n, const TowerParams &tp, int grid, const TowerGeom &g, hipStream_t st) {
    if constexpr (NT <= 3) {
        if (g.waves == 8) return g.ck == 8 ? launch_tower<NT, 8, 8>(n, tp, grid, g.lds, st) : launch_tower<NT, 4, 8>(n, tp, grid, g.lds, st);
    }
    return g.ck == 8 ? launch_tower<NT, 8, 4>(n, tp, grid, g.lds, st) : launch_tower<NT, 4, 4>(n, tp, grid, g.lds, st);
}

template <int NT, bool RP1, int R3> static hipError_t launch_x3_r3(const az_net *n, const TowerParams &tp, int grid, int lds, hipStream_t st) {
    static bool attr_set[AZ_MAX_DEVICES] = {false};
    const int dv = n->d.device;
    if (dv < 0 || dv >= AZ_MAX_DEVICES || !attr_set[dv]) {
        hipError_t s = hipFuncSetAttribute((const void *)az_tower_x3_kernel<NT, 4, RP1, R3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (s != hipSuccess) return s;
        if (dv >= 0 && dv < AZ_MAX_DEVICES) attr_set[dv] = true;
    }
    hipLaunchKernelGGL((az_tower_x3_kernel<NT, 4, RP1, R3>), dim3(grid), dim3(256), lds, st, tp);
    return hipGetLastError();
}
template <int NT, bool RP1> static hipError_t launch_x3(const az_net *n, const TowerParams &tp, int grid, int lds, hipStream_t st) {
    return n->r3 == 2 ? launch_x3_r3<NT, RP1, 2>(n, tp, grid, lds, st) : launch_x3_r3<NT, RP1, 16>(n, tp, grid, lds, st);
}

template <bool X3> static int launch_head(az_net *n, const HeadParams &hp, int n_boards, hipStream_t st) {
    if (n->n_ot > OTG) { // large action space: logits over (board tile x output-tile group), then softmax
        constexpr int lds_logits = HEAD_RING * 4 * HEAD_OTG * 1024; // RING chunks of 16 KiB
        const int dv = n->d.device;
        const int col_groups = (n->n_ot + HEAD_OTG - 1) / HEAD_OTG;
        // one board tile per wave (HEAD_MT = 1; two measured the same, tools/net_microbench.py): 64-KiB workgroups, two per CU
        static bool lg_attr[AZ_MAX_DEVICES] = {false};
        if (dv < 0 || dv >= AZ_MAX_DEVICES || !lg_attr[dv]) {
            NCHK(n, hipFuncSetAttribute((const void *)az_head_logits_kernel<X3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            if (dv >= 0 && dv < AZ_MAX_DEVICES) lg_attr[dv] = true;
        }
        hipLaunchKernelGGL((az_head_logits_kernel<X3, 1>), dim3(((n_boards + 63) / 64 + 7) / 8 * 8 * col_groups), dim3(256), lds_logits, st, hp, n->logits);
        hipLaunchKernelGGL(az_head_softmax_kernel<X3>, dim3((n_boards + 3) / 4), dim3(256), 0, st, hp, (const float *)n->logits);
    } else {
        static bool head_attr[AZ_MAX_DEVICES] = {false};
        const int dv = n->d.device;
        if (dv < 0 || dv >= AZ_MAX_DEVICES || !head_attr[dv]) {
            NCHK(n, hipFuncSetAttribute((const void *)az_head_kernel<X3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            if (dv >= 0 && dv < AZ_MAX_DEVICES) head_attr[dv] = true;
        }
        hipLaunchKernelGGL(az_head_kernel<X3>, dim3((n_boards + 15) / 16), dim3(HEAD_NW * 64), n->lds_head, st, hp);
    }
    NCHK(n, hipGetLastError());
    return AZ_OK;
}

extern "C" int az_net_forward(az_net *n, const float *obs, float *priors, float *values, int32_t n_boards, void *stream) {
    if (!n || !obs || !priors || !values || n_boards < 1) return AZ_E_INVALID;
    if (n_boards > n->max_boards) {
        n->err = "n_boards exceeds az_net_reserve()";
        return AZ_E_STATE;
    }
    NCHK(n, hipSetDevice(n->d.device)); // the launch must pair `stream` with the device the net lives on
    hipStream_t st = (hipStream_t)stream;
    HeadParams hp;
    hp.HW = n->d.rows * n->d.cols;
    hp.A = n->d.num_actions;
    hp.n_ot = n->n_ot;
    hp.ksteps = hp.HW * AZ_NET_XOUT_C / 32;
    hp.n_boards = n_boards;
    hp.x = n->xout;
    hp.x_lo = n->xout_lo;
    hp.fc_w = n->fc_w;
    hp.fc_w_lo = n->fc_w_lo;
    hp.fc_b = n->fc_b;
    hp.priors = priors;
    hp.values = values;
    if (n->precision == AZ_NET_PREC_F16X3) { // split-fp16 tower: one board per wave, 4 waves per workgroup
        const X3Geom g = x3_geom(n->d.rows, n->d.cols, n->r3);
        TowerParams tp;
        tp.H = n->d.rows;
        tp.W = n->d.cols;
        tp.HW = tp.H * tp.W;
        tp.cells = g.cells;
        tp.rs = g.rs;
        tp.tpb = g.tpb;
        tp.off_epi = g.off_epi;
        tp.off_act = g.off_act;
        tp.cin = n->d.in_planes;
        tp.n_convs = 2 * n->d.n_blocks;
        tp.n_boards = n_boards;
        tp.bpw = 1;
        tp.rcells = g.rcells;
        tp.zcell = g.zcell;
        tp.conv_w = n->conv_w;
        tp.epi = n->epi;
        tp.skip_w = n->skip_w;
        memcpy(tp.in_scale, n->in_affine, 32);
        memcpy(tp.in_shift, n->in_affine + 8, 32);
        tp.obs = obs;
        tp.xout = n->xout;
        tp.xout_lo = n->xout_lo;
        const int grid = (n_boards + 3) / 4;
        hipError_t s;
        if (g.nt <= 3) s = g.rp1 ? launch_x3<3, true>(n, tp, grid, g.lds, st) : launch_x3<3, false>(n, tp, grid, g.lds, st);
        else s = launch_x3<4, false>(n, tp, grid, g.lds, st);
        if (s != hipSuccess) {
            n->err = std::string("f16x3 tower launch: ") + hipGetErrorString(s);
            return AZ_E_HIP;
        }
        return launch_head<true>(n, hp, n_boards, st);
    }
    // boards per wave: one workgroup (4 waves) per CU is resident, a launch runs in ceil(WGs / 256) rounds and a
    // round costs ~ (column tiles + fixed part): pick the bpw that minimises rounds x tiles for THIS batch size.
    // Work partition for THIS batch size.  Candidates: boards per wave x {4, 8} waves per workgroup.  A launch runs in
    // ceil(WGs / resident WGs) rounds; a round costs ~ (2*tiles + 1), x1.5 when two waves share each SIMD.
    TowerGeom g = tower_geom(1, 4, n->d.rows, n->d.cols);
    {
        double best_cost = -1;
        for (int bpw = 1; bpw <= n->bpw_max; bpw++)
            for (int waves = 4; waves <= 8; waves += 4) {
                TowerGeom c = tower_geom(bpw, waves, n->d.rows, n->d.cols);
                if (c.lds > 160 * 1024 || (waves == 8 && c.nt > 3)) continue; // 8 waves need <= 256 registers each
                long wgs = (n_boards + waves * bpw - 1) / (waves * bpw);
                int per_cu = 1; // every variant's registers/LDS admit one workgroup per CU
                long rounds = (wgs + 256 * per_cu - 1) / (256 * per_cu);
                bool two_per_simd = waves == 8 || (per_cu == 2 && wgs > 256);
                double cost = rounds * (2.0 * (c.nt < 3 ? 3 : c.nt) + 1.0) * (two_per_simd ? 1.5 : 1.0);
                if (best_cost < 0 || cost < best_cost) {
                    best_cost = cost;
                    g = c;
                }
            }
    }
    TowerParams tp;
    tp.H = n->d.rows;
    tp.W = n->d.cols;
    tp.HW = tp.H * tp.W;
    tp.cells = g.cells;
    tp.rs = g.rs;
    tp.tpb = g.tpb;
    tp.off_epi = g.off_epi;
    tp.cin = n->d.in_planes;
    tp.n_convs = 2 * n->d.n_blocks;
    tp.n_boards = n_boards;
    tp.bpw = g.bpw;
    tp.rcells = g.rcells;
    tp.zcell = g.zcell;
    tp.off_act = g.off_act;
    tp.conv_w = n->conv_w;
    tp.epi = n->epi;
    tp.skip_w = n->skip_w;
    memcpy(tp.in_scale, n->in_affine, 32);
    memcpy(tp.in_shift, n->in_affine + 8, 32);
    tp.obs = obs;
    tp.xout = n->xout;
    tp.xout_lo = nullptr;
    int per_wg = g.waves * g.bpw, grid = (n_boards + per_wg - 1) / per_wg;
    hipError_t s;
    switch (g.nt < 3 ? 3 : g.nt) {
    case 3: s = launch_tower_ck<3>(n, tp, grid, g, st); break;
    default: s = launch_tower_ck<4>(n, tp, grid, g, st); break;
    }
    if (s != hipSuccess) {
        n->err = std::string("tower launch: ") + hipGetErrorString(s);
        return AZ_E_HIP;
    }
    return launch_head<false>(n, hp, n_boards, st);
}

extern "C" int az_net_read_tower(az_net *n, float *out, int32_t n_boards) {
    if (!n || !out || n_boards < 1 || n_boards > n->max_boards) return AZ_E_INVALID;
    NCHK(n, hipSetDevice(n->d.device));
    NCHK(n, hipDeviceSynchronize());
    size_t cnt = (size_t)n_boards * n->d.rows * n->d.cols * AZ_NET_XOUT_C;
    std::vector<_Float16> h(cnt);
    NCHK(n, hipMemcpy(h.data(), n->xout, cnt * 2, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < cnt; i++) out[i] = (float)h[i];
    if (n->precision == AZ_NET_PREC_F16X3) { // hi + lo / 2048
        NCHK(n, hipMemcpy(h.data(), n->xout_lo, cnt * 2, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < cnt; i++) out[i] += (float)h[i] * (1.0f / 2048.0f);
    }
    return AZ_OK;
}
