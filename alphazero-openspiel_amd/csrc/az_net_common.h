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
