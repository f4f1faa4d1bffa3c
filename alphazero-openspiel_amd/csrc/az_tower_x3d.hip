// az_tower_x3d.hip — instantiations and launcher of az_tower_x3d_kernel (az_tower_x3d.h): the fp32-grade tower on packed column tiles.
#include "az_head_params.h"
#include "az_tower_x3d.h"

template <int V> static hipError_t launch_x3d(int device, const TowerParams &tp, int grid, hipStream_t st) {
    static bool attr_set[AZ_MAX_DEVICES] = {false};
    static_assert(X3DV<V>::NTILES == az_x3d_variant(V).tiles && X3DV<V>::PC == az_x3d_variant(V).pc && X3DV<V>::RS == az_x3d_variant(V).rs && X3DV<V>::R == az_x3d_variant(V).R,
                  "host table (az_head_params.h) and kernel variants agree");
    if (device < 0 || device >= AZ_MAX_DEVICES || !attr_set[device]) {
        hipError_t s = hipFuncSetAttribute((const void *)az_tower_x3d_kernel<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (s != hipSuccess) return s;
        if (device >= 0 && device < AZ_MAX_DEVICES) attr_set[device] = true;
    }
    constexpr int lds = X3DG<V>::LDS;
    hipLaunchKernelGGL((az_tower_x3d_kernel<V>), dim3(grid), dim3(512), lds, st, tp);
    return hipGetLastError();
}
// variant: 0 = 8 boards in 18 tiles (6x6), 1 = 4 boards in 16 tiles (8x8); the 6x6 roles spill ~60 registers outside the k-loop and still gain
hipError_t az_launch_tower_x3d(int device, int variant, const TowerParams &tp, int grid, hipStream_t st) {
    if (variant == 0) return launch_x3d<0>(device, tp, grid, st);
    if (variant == 1) return launch_x3d<1>(device, tp, grid, st);
    if (variant == 2) return launch_x3d<2>(device, tp, grid, st);
    return hipErrorInvalidValue;
}
