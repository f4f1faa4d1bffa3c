// az_net_common.h — types, LDS access helpers and launch parameters shared by the PV-net kernels (included by az_net.hip only;
// NOT part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

#include "../../include/az_net.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define OCT_B 16 // one cell of one channel-octet plane: 8 fp16
#define AZ_NET_K0STEPS 4 // k-steps of conv 0 on the device (9 taps x the one octet holding the input planes, padded to 16 groups)
#define N_OCT 7  // 56 channels
#define AZ_MAX_DEVICES 64
#define X3_LOFF_RP1 (N_OCT * 96 * OCT_B) // f16x3, row-pair boards (W <= 7, H <= 6): the lo planes sit a compile-time distance after hi

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
    int xout_c;        // channel stride of xout: AZ_NET_XOUT_C, or 52 where fc1 is az_head_gemm_kernel (az_net_create)
    // az_tower_x3c_kernel only: fc1 + softmax + tanh inside the launch when fc_w is set (one output tile: A + 1 <= 16)
    const _Float16 *fc_w, *fc_w_lo; // [ksteps][64][8] (az_head_params.h)
    const float *fc_b;
    float *priors, *values;
    int A, fc_ksteps;
    // az_tower_x3d_kernel only (az_tower_x3d.h): packed, permuted column tiles of xd_nb boards per workgroup
    int xd_nb, xd_R, xd_rs;       // boards per workgroup; cells per board region; row stride (cell = b R + (y + 1) rs + x + 1)
    const uint16_t *xd_pos;       // [16 x tiles] column -> board << 8 | position
    const uint16_t *xd_sdst;      // [16 x tiles][8] column, tap plane t -> the column that takes this column's tile-X term (0xFFFF: off the board)
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
// An untracked asynchronous read looks "defined" to the compiler the moment it is issued, so a LONG-LIVED destination may be
// copied (e.g. parked in an AGPR) before its data has arrived.  launder(x), placed after the s_waitcnt that covers the read,
// makes every later use depend on an asm that runs after the wait (volatile asms keep their order).
__device__ __forceinline__ void launder(half8 &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void launder(f32x4 &v) { asm volatile("" : "+v"(v)); }
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void launder(f32x2 &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void launder_u(unsigned &v) { asm volatile("" : "+v"(v)); }
// A kernel whose accumulators live in AGPRs (more than 256 registers per lane): the value an MFMA leaves there is copied to a
// VGPR for the epilogue's vector arithmetic, and the register allocator puts that copy right behind the MFMA - in the last k-step
// every MFMA is then followed by the wait for its own result.  pin_acc(x), placed behind the k-loop, keeps x in its AGPR until there.
__device__ __forceinline__ void pin_acc(f32x4 &v) { asm volatile("" : "+a"(v)); }
// (x0, x1) -> the packed fp16 pairs (hi0, hi1) and (lo0, lo1) of the split-fp16 format: hi = fp16(x) (round to nearest even),
// lo = fp16((x - hi) * 2048), computed as fma(hi, -2048, x * 2048) - x * 2048 and the fma's result are exact (hi is x rounded to
// 11 bits: the difference has at most 13), so these are the bits of the subtract-then-scale form.  Five instructions for two
// values (v_cvt_pk_f16_f32, two multiplies, v_fma_mixlo / mixhi_f16 reading the fp16 operand in place and writing the fp16
// result directly) against the twelve hipcc emits for the plain C++ expression.
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned &hi, unsigned &lo) {
    const float m = -2048.0f, t0 = x0 * 2048.0f, t1 = x1 * 2048.0f;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(x0), "v"(x1));
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=&v"(lo) : "v"(hi), "v"(m), "v"(t0));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(m), "v"(t1));
}
// BETWEEN the convs of a tower the lo half is kept UNSCALED: lo0 = fp16(x - hi) (one v_fma_mix per value: fma(hi, -1, x), exact
// before the conversion).  The weights carry the factor instead - both halves of a weight are stored x 2048 (hi' = 2048 fp16(w),
// lo = 2048 (w - fp16(w)): az_net.hip) - so that the three products of a split-fp16 multiply, hi' bh + hi' bl0 + lo bh, all come
// out 2048 times the true product and go into ONE fp32 accumulator (no second accumulator set for the cross terms: a third of the
// accumulator registers).  An unscaled lo is a subnormal fp16 for |x| < 0.125: its absolute error is then 2^-25 at most, which a
// 450-term conv with |w| ~ 0.05 turns into ~2e-8 of its output - below fp32's own rounding of the sum.
__device__ __forceinline__ void split_pair_planes(float x0, float x1, unsigned &hi, unsigned &lo0) {
    const float m = -1.0f;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(x0), "v"(x1));
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=&v"(lo0) : "v"(hi), "v"(m), "v"(x0));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo0) : "v"(hi), "v"(m), "v"(x1));
}
__device__ __forceinline__ void split4_planes(const f32x4 &v, half4 &hi, half4 &lo0) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    unsigned h0, h1, l0, l1;
    split_pair_planes(v[0], v[1], h0, l0);
    split_pair_planes(v[2], v[3], h1, l1);
    hi = __builtin_bit_cast(half4, (u32x2){h0, h1});
    lo0 = __builtin_bit_cast(half4, (u32x2){l0, l1});
}
#define X3_WSCALE 2048.0f // the factor the fp32-grade towers' weights (and therefore their accumulators) carry
// four values at once (the fp32-grade towers' epilogues)
__device__ __forceinline__ void split4_f16x3(const f32x4 &v, half4 &hi, half4 &lo) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    unsigned h0, h1, l0, l1;
    split_pair(v[0], v[1], h0, l0);
    split_pair(v[2], v[3], h1, l1);
    hi = __builtin_bit_cast(half4, (u32x2){h0, h1});
    lo = __builtin_bit_cast(half4, (u32x2){l0, l1});
}
// a + a2 / 2048 in one rounding: a2 / 2048 is exact (a power of two), so fma(a2, 1/2048, a) has the bits of multiply-then-add
__device__ __forceinline__ f32x4 comb_f16x3(const f32x4 &a, const f32x4 &a2) {
    const f32x4 is = {1.0f / 2048.0f, 1.0f / 2048.0f, 1.0f / 2048.0f, 1.0f / 2048.0f};
    return __builtin_elementwise_fma(a2, is, a);
}
// an address sum the compiler must not hoist out of the conv loop (it is invariant there: hoisted, every (tile, k-step) pair of
// az_tower_x3d_kernel would hold a register for the whole kernel)
__device__ __forceinline__ unsigned addr_add(unsigned a, unsigned b) {
    unsigned r;
    asm volatile("v_add_u32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// untracked 8-byte LDS accesses (volatile asms keep their program order; LDS executes one wave's operations in order)
template <int OFF> __device__ __forceinline__ void lds_read64_off(f32x2 &dst, unsigned lds_byte_addr) {
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_byte_addr), "n"(OFF));
}
__device__ __forceinline__ void lds_write64(unsigned lds_byte_addr, f32x2 v) { asm volatile("ds_write_b64 %0, %1" ::"v"(lds_byte_addr), "v"(v) : "memory"); }
template <int OFF> __device__ __forceinline__ void lds_write64_off(unsigned lds_byte_addr, f32x2 v) {
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(lds_byte_addr), "v"(v), "n"(OFF) : "memory");
}
#define READ_A(dst, addr, off) lds_read128_off<(off)>(dst, addr)
template <int OFF> __device__ __forceinline__ void lds_read32_off(unsigned &dst, unsigned lds_byte_addr) {
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_byte_addr), "n"(OFF));
}
#define READ_B32_OFF(dst, addr, off) lds_read32_off<(off)>(dst, addr)
#define READ_B(dst, addr) lds_read128(dst, addr)
#define READ_B_OFF(dst, addr, off) lds_read128_off<(off)>(dst, addr)
// Geometry of az_tower_x3b_kernel (az_tower_x3b.h): weight stream, LDS map.  Shared with the host packing in az_net.hip.
// chunk = 4 k-steps of records (6 KiB each: hi mt 0..2, lo mt 0..2), then the extra fragments of its k-steps:
//   part 1 (k-steps 4..7): T, Xhi, Xlo of k-step 6, then of k-step 7;  part 3 (k-steps 12..14): T of the gather k-step
constexpr int x3b_part_bytes(int part) { return part == 1 ? 4 * 6144 + 6 * 1024 : part == 3 ? 3 * 6144 + 1024 : 4 * 6144; }
constexpr int x3b_part_off(int part) { return part == 0 ? 0 : x3b_part_off(part - 1) + x3b_part_bytes(part - 1); }
struct X3B {
    static constexpr int FR = 1024;          // one A fragment: 64 lanes x 16 B
    static constexpr int REC2 = 6 * FR;      // one k-step of tiles 0..2: hi mt 0..2, lo mt 0..2
    static constexpr int CK = 4, NKS = 15, PARTS = 4;
    static constexpr int part_bytes(int part) { return x3b_part_bytes(part); }
    static constexpr int part_off(int part) { return x3b_part_off(part); }
    static constexpr int CONV_B = x3b_part_off(3) + x3b_part_bytes(3);
    static constexpr int C0_B = AZ_NET_K0STEPS * REC2 + AZ_NET_K0STEPS * FR; // conv 0: 4 k-steps of records + T of each
    static constexpr int CHUNK_S = x3b_part_bytes(1);                            // LDS stride of the two chunk buffers
    static constexpr int S_PLANE = 48 * 8, S_WAVE = 9 * S_PLANE;             // scratch: [plane][y * 8 + x][2 channels] fp32
    static constexpr int PLANE_B = 96 * OCT_B;                               // one channel-octet plane (rcells = 96)
    static constexpr int LO_OFF = 6 * PLANE_B + 96 * 4;                      // hi planes: 6 octets + compact plane; lo follows
    static constexpr int OFF_EPI = 2 * CHUNK_S;
    static constexpr int OFF_S = OFF_EPI + 2048 + 256 * 16; // epilogue ring (2 KiB) + trash slots (16 B per thread)
    static constexpr int OFF_ACT = OFF_S + 4 * S_WAVE;
    static constexpr int LDS = OFF_ACT + 4 * 2 * LO_OFF;
    static constexpr int tap_of_plane(int t) { return t < 4 ? t : t + 1; } // planes 0..7 = taps 0,1,2,3,5,6,7,8 (plane 8 = tap 4)
};
static_assert(X3B::LDS <= 160 * 1024, "x3b LDS budget");
static_assert(X3B::C0_B <= X3B::CHUNK_S, "conv 0 must fit a chunk buffer");

// Geometry of az_tower_x3d_kernel (az_tower_x3d.h): the weight stream in 3-k-step chunks (two 18-KiB LDS buffers leave room for the
// planes of eight connect_four boards), the fixed part of the LDS map.  Shared with the host packing in az_net.hip.
//   conv 0:      part 0: k-steps 0, 1, then T of each | part 1: k-steps 2, 3, then T of each
//   conv c >= 1: parts [0,3) [3,6) [6,8) + (T, Xhi, Xlo) of k-steps 6, 7 | [8,11) [11,14) [14] + T of the gather k-step
constexpr int x3d_part_ks0(int part) { return part == 0 ? 0 : part == 1 ? 3 : part == 2 ? 6 : part == 3 ? 8 : part == 4 ? 11 : 14; }
constexpr int x3d_part_len(int part) { return part == 2 ? 2 : part == 5 ? 1 : 3; }
constexpr int x3d_part_bytes(int part) { return x3d_part_len(part) * 6144 + (part == 2 ? 6 * 1024 : part == 5 ? 1024 : 0); }
constexpr int x3d_part_off(int part) { return part == 0 ? 0 : x3d_part_off(part - 1) + x3d_part_bytes(part - 1); }
struct X3D {
    static constexpr int FR = 1024, REC2 = 6 * FR, NKS = 15, PARTS = 6, PARTS0 = 2;
    static constexpr int part_ks0(int part) { return x3d_part_ks0(part); }
    static constexpr int part_len(int part) { return x3d_part_len(part); }
    static constexpr int part_of(int ks) { return ks < 3 ? 0 : ks < 6 ? 1 : ks < 8 ? 2 : ks < 11 ? 3 : ks < 14 ? 4 : 5; }
    static constexpr int part_bytes(int part) { return x3d_part_bytes(part); }
    static constexpr int part_off(int part) { return x3d_part_off(part); }
    static constexpr int CONV_B = x3d_part_off(5) + x3d_part_bytes(5);
    static constexpr int C0_PART_B = 2 * REC2 + 2 * FR, C0_B = 2 * C0_PART_B;
    static constexpr int CHUNK_S = 3 * REC2;            // LDS stride of the two chunk buffers (the largest part)
    static constexpr int OFF_EPI = 2 * CHUNK_S;         // epilogue ring (2 KiB) + trash slots (16 B per thread)
    static constexpr int OFF_S = OFF_EPI + 2048 + 256 * 16; // scratch: [9 tap planes][columns][2 channels] fp32
    static constexpr int tap_of_plane(int t) { return t < 4 ? t : t + 1; }
};
static_assert(x3d_part_bytes(2) <= X3D::CHUNK_S && X3D::C0_PART_B <= X3D::CHUNK_S && X3D::CONV_B == 15 * 6144 + 7 * 1024, "every part fits a chunk buffer");

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{})
template <class F, int... I> __device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// Per-lane address tables of one wave of a tower kernel: where its NT*16 columns live in the LDS image (pos_addr, and p6_addr
// in the compact plane of channels 48, 49), which global (board, position) each column is (grow, -1 = padding), and the byte
// offset (tap shift + octet plane) of the lane's k-group in every k-step (koff: the 15/16-k-step convs, ksp: the four taps
// of the gather k-step, koff0: conv 0).  RP1: row-pair tiles, one board per wave: the offsets carry the FULL LDS address of
// tile 0's fragment, tile nt is an immediate away.
template <int NT, bool RP1, bool L15> struct TowerTables {
    int pos_addr[NT], grow[NT], p6_addr[NT];
    int koff[AZ_NET_KSTEPS], ksp[4], koff0[AZ_NET_K0STEPS];
    __device__ __forceinline__ void init(const TowerParams &p, int region, int plane_b, unsigned lds_base, int board0, int q, int l15) {
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
        // gather reads of the last k-step touch 16 consecutive dwords per 16 columns (at the octet planes' 16-byte cell
        // stride they were 4-way bank conflicted: 17 % of the f16 kernel's LDS cycles before, 11 % after)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) p6_addr[nt] = region + 6 * plane_b + ((pos_addr[nt] - region) >> 2);
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
#pragma unroll
        for (int ks = 0; ks < AZ_NET_K0STEPS; ks++) {
            int g = 4 * ks + q;
            int dy = g / 3 - 1, dx = g - (g / 3) * 3 - 1;
            koff0[ks] = g < 9 ? (dy * p.rs + dx) * OCT_B : 0;
            if (RP1) koff0[ks] += (int)lds_base + pos_addr[0];
        }
    }
};
