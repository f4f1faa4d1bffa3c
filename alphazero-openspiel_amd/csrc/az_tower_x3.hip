// az_tower_x3.hip — instantiations and launcher of az_tower_x3_kernel (az_tower_x3.h): split-fp16 operands, fp32 grade.
#include "az_head_params.h"
#include "az_tower_f16.h" // WRec
#include "az_tower_x3.h"
#include "az_tower_x3b.h"
#include "az_tower_x3c.h"

template <int NT, bool RP1, int R3> static hipError_t launch_r3(int dv, const TowerParams &tp, int grid, int lds, hipStream_t st) {
    static bool attr_set[AZ_MAX_DEVICES] = {false};
    if (dv < 0 || dv >= AZ_MAX_DEVICES || !attr_set[dv]) {
        hipError_t s = hipFuncSetAttribute((const void *)az_tower_x3_kernel<NT, 4, RP1, R3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (s != hipSuccess) return s;
        if (dv >= 0 && dv < AZ_MAX_DEVICES) attr_set[dv] = true;
    }
    hipLaunchKernelGGL((az_tower_x3_kernel<NT, 4, RP1, R3>), dim3(grid), dim3(256), lds, st, tp);
    return hipGetLastError();
}
template <int NT, bool RP1> static hipError_t launch(int dv, int r3, const TowerParams &tp, int grid, int lds, hipStream_t st) {
    return r3 == 2 ? launch_r3<NT, RP1, 2>(dv, tp, grid, lds, st) : launch_r3<NT, RP1, 16>(dv, tp, grid, lds, st);
}
hipError_t az_launch_tower_x3(int device, int nt, bool rp1, int r3, const TowerParams &tp, int grid, int lds, hipStream_t st) {
    if (nt <= 3) return rp1 ? launch<3, true>(device, r3, tp, grid, lds, st) : launch<3, false>(device, r3, tp, grid, lds, st);
    return launch<4, false>(device, r3, tp, grid, lds, st);
}

hipError_t az_launch_tower_x3b(int device, const TowerParams &tp, int grid, hipStream_t st) {
    static bool attr_set[AZ_MAX_DEVICES] = {false};
    if (device < 0 || device >= AZ_MAX_DEVICES || !attr_set[device]) {
        hipError_t s = hipFuncSetAttribute((const void *)az_tower_x3b_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (s != hipSuccess) return s;
        if (device >= 0 && device < AZ_MAX_DEVICES) attr_set[device] = true;
    }
    hipLaunchKernelGGL((az_tower_x3b_kernel<3>), dim3(grid), dim3(256), X3B::LDS, st, tp);
    return hipGetLastError();
}

template <int BPW> static hipError_t launch_x3c(int device, const TowerParams &tp, int n_boards, hipStream_t st) {
    static bool attr_set[AZ_MAX_DEVICES] = {false};
    if (device < 0 || device >= AZ_MAX_DEVICES || !attr_set[device]) {
        hipError_t s = hipFuncSetAttribute((const void *)az_tower_x3c_kernel<3, BPW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (s != hipSuccess) return s;
        if (device >= 0 && device < AZ_MAX_DEVICES) attr_set[device] = true;
    }
    hipLaunchKernelGGL((az_tower_x3c_kernel<3, BPW>), dim3((n_boards + BPW - 1) / BPW), dim3(256 * BPW), X3B::LDS, st, tp);
    return hipGetLastError();
}
hipError_t az_launch_tower_x3c(int device, int bpw, const TowerParams &tp, int n_boards, hipStream_t st) {
    return bpw == 2 ? launch_x3c<2>(device, tp, n_boards, st) : launch_x3c<1>(device, tp, n_boards, st);
}
