// az_head_params.h — launch parameters of the head kernels and the launcher prototypes of every kernel translation unit
// (az_tower_f16.hip, az_tower_x3.hip, az_head.hip).  Internal; NOT part of the C ABI.
#pragma once
#include "az_net_common.h"

struct HeadParams {
    int HW, A, n_ot, ksteps, n_boards;
    int K;                // halves per board row of x: HW * (channel stride of the tower output); ksteps = ceil(K / 32)
    const _Float16 *x;    // [B][K]
    const _Float16 *fc_w; // [n_ot][ksteps][64][8]
    const _Float16 *x_lo, *fc_w_lo; // f16x3: the lo halves (scaled by 2048), same layouts
    const float *fc_b;
    float *priors, *values;
};

#define OTG 8       // output tiles per pass of az_head_kernel; action spaces with more tiles take az_head_gemm_kernel + az_head_softmax_kernel
#define HEAD_NW 8   // waves per workgroup of az_head_kernel: the K reduction is split over them

// Launchers (one per translation unit, so the kernel families compile side by side).  `device` indexes the per-device
// "dynamic LDS attribute set" flags; every launcher returns hipGetLastError() of its launch.
hipError_t az_launch_tower_f16(int device, int nt, int ck, int waves, int r3, const TowerParams &tp, int grid, int lds, hipStream_t st);
hipError_t az_launch_tower_f16c(int device, int ck, const TowerParams &tp, int n_boards, int lds, hipStream_t st); // one board per workgroup (small batches); ck: k-steps per weight chunk, 8 or 4
hipError_t az_launch_tower_x3(int device, int nt, bool rp1, int r3, const TowerParams &tp, int grid, int lds, hipStream_t st);
hipError_t az_launch_tower_x3b(int device, const TowerParams &tp, int grid, hipStream_t st); // row-pair boards, <= 50 filters
// packed column tiles (az_tower_x3d.h); variant: 0 = 8 boards in 18 tiles (6x6), 1 = 4 boards in 16 tiles (8x8), 2 = 8 boards in 21 tiles (6x7)
hipError_t az_launch_tower_x3d(int device, int variant, const TowerParams &tp, int grid, hipStream_t st);
struct X3DVariant { int pc, tiles, rs, R; }; // cells per plane, column tiles per workgroup, row stride, cells per board region
constexpr int AZ_X3D_VARIANTS = 3;
constexpr X3DVariant az_x3d_variant(int v) { return v == 0 ? X3DVariant{480, 18, 8, 58} : v == 1 ? X3DVariant{416, 16, 9, 84} : X3DVariant{480, 21, 8, 57}; }
hipError_t az_launch_tower_x3c(int device, int bpw, const TowerParams &tp, int n_boards, hipStream_t st); // the same, a board per four waves, bpw boards per workgroup
hipError_t az_launch_head(int device, bool x3, const HeadParams &hp, int n_boards, int lds_head, float *logits, hipStream_t st);
