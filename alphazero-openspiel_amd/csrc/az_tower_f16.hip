// az_tower_f16.hip — instantiations and launcher of az_tower_kernel (az_tower_f16.h): fp16 MFMA operands.
#include "az_head_params.h"
#include "az_tower_f16.h"
#include "az_tower_f16c.h"

template <int NT, int CK, int WAVES, bool RP1, int R3> static hipError_t launch_r3(int dv, const TowerParams &tp, int grid, int lds, hipStream_t st) {
    static bool attr_set[AZ_MAX_DEVICES] = {false}; // the attribute is per (function, device)
    if (dv < 0 || dv >= AZ_MAX_DEVICES || !attr_set[dv]) {
        hipError_t s = hipFuncSetAttribute((const void *)az_tower_kernel<NT, CK, WAVES, RP1, R3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (s != hipSuccess) return s;
        if (dv >= 0 && dv < AZ_MAX_DEVICES) attr_set[dv] = true;
    }
    hipLaunchKernelGGL((az_tower_kernel<NT, CK, WAVES, RP1, R3>), dim3(grid), dim3(WAVES * 64), lds, st, tp);
    return hipGetLastError();
}
template <int NT, int CK, int WAVES, bool RP1> static hipError_t launch_rp(int dv, int r3, const TowerParams &tp, int grid, int lds, hipStream_t st) {
    return r3 == 2 ? launch_r3<NT, CK, WAVES, RP1, 2>(dv, tp, grid, lds, st) : launch_r3<NT, CK, WAVES, RP1, 16>(dv, tp, grid, lds, st);
}
template <int NT, int CK, int WAVES> static hipError_t launch(int dv, int r3, const TowerParams &tp, int grid, int lds, hipStream_t st) {
    // row-pair tiles (row stride 8), one board per wave, every tile of the wave on that board
    if (tp.tpb && tp.bpw == 1 && tp.rs == 8 && tp.tpb <= NT) return launch_rp<NT, CK, WAVES, true>(dv, r3, tp, grid, lds, st);
    return launch_rp<NT, CK, WAVES, false>(dv, r3, tp, grid, lds, st);
}
template <int NT> static hipError_t launch_ck(int dv, int ck, int waves, int r3, const TowerParams &tp, int grid, int lds, hipStream_t st) {
    if constexpr (NT <= 3) { // (NT = 4 with 8 waves spills 42 registers under the 256 limit: 373 vs 351 us on 2048 8x8 boards)
        if (waves == 8) return ck == 8 ? launch<NT, 8, 8>(dv, r3, tp, grid, lds, st) : launch<NT, 4, 8>(dv, r3, tp, grid, lds, st);
    }
    return ck == 8 ? launch<NT, 8, 4>(dv, r3, tp, grid, lds, st) : launch<NT, 4, 4>(dv, r3, tp, grid, lds, st);
}
hipError_t az_launch_tower_f16(int device, int nt, int ck, int waves, int r3, const TowerParams &tp, int grid, int lds, hipStream_t st) {
    return nt <= 3 ? launch_ck<3>(device, ck, waves, r3, tp, grid, lds, st) : launch_ck<4>(device, ck, waves, r3, tp, grid, lds, st);
}

// small batches of a row-pair board with <= 50 filters: one board per four-wave workgroup (az_tower_f16c.h)
template <int CK, int RING> static hipError_t launch_f16c(int dv, const TowerParams &tp, int n_boards, int lds, hipStream_t st) {
    static bool attr_set[AZ_MAX_DEVICES] = {false};
    if (dv < 0 || dv >= AZ_MAX_DEVICES || !attr_set[dv]) {
        hipError_t s = hipFuncSetAttribute((const void *)az_tower_f16c_kernel<3, CK, RING>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (s != hipSuccess) return s;
        if (dv >= 0 && dv < AZ_MAX_DEVICES) attr_set[dv] = true;
    }
    hipLaunchKernelGGL((az_tower_f16c_kernel<3, CK, RING>), dim3(n_boards), dim3(256), lds, st, tp);
    return hipGetLastError();
}
hipError_t az_launch_tower_f16c(int dv, int ck, const TowerParams &tp, int n_boards, int lds, hipStream_t st) {
    return ck == 8 ? launch_f16c<8, 4>(dv, tp, n_boards, lds, st) : launch_f16c<4, 3>(dv, tp, n_boards, lds, st); // (ring depths: az_net.hip's LDS layout)
}
