// az_head.hip — instantiations and launcher of the head kernels (az_head.h): fc1 + softmax + tanh.
#include "az_head_params.h"
#include "az_head.h"

template <bool X3> static hipError_t launch(int dv, const HeadParams &hp, int n_boards, int lds_head, float *logits, hipStream_t st) {
    hipError_t s;
    if (hp.n_ot > OTG) { // large action space: logits over (board tile x output-tile group), then softmax
        constexpr int lds_logits = HEAD_RING * 4 * HEAD_OTG * 1024; // RING chunks of 16 KiB
        const int col_groups = (hp.n_ot + HEAD_OTG - 1) / HEAD_OTG;
        // one board tile per wave (HEAD_MT = 1; two measured the same, tools/net_microbench.py): 64-KiB workgroups, two per CU
        static bool lg_attr[AZ_MAX_DEVICES] = {false};
        if (dv < 0 || dv >= AZ_MAX_DEVICES || !lg_attr[dv]) {
            s = hipFuncSetAttribute((const void *)az_head_logits_kernel<X3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (s != hipSuccess) return s;
            if (dv >= 0 && dv < AZ_MAX_DEVICES) lg_attr[dv] = true;
        }
        hipLaunchKernelGGL((az_head_logits_kernel<X3, 1>), dim3(((n_boards + 63) / 64 + 7) / 8 * 8 * col_groups), dim3(256), lds_logits, st, hp, logits);
        hipLaunchKernelGGL(az_head_softmax_kernel<X3>, dim3((n_boards + 3) / 4), dim3(256), 0, st, hp, (const float *)logits);
    } else {
        static bool head_attr[AZ_MAX_DEVICES] = {false};
        if (dv < 0 || dv >= AZ_MAX_DEVICES || !head_attr[dv]) {
            s = hipFuncSetAttribute((const void *)az_head_kernel<X3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (s != hipSuccess) return s;
            if (dv >= 0 && dv < AZ_MAX_DEVICES) head_attr[dv] = true;
        }
        hipLaunchKernelGGL(az_head_kernel<X3>, dim3((n_boards + 15) / 16), dim3(HEAD_NW * 64), lds_head, st, hp);
    }
    return hipGetLastError();
}
hipError_t az_launch_head(int device, bool x3, const HeadParams &hp, int n_boards, int lds_head, float *logits, hipStream_t st) {
    return x3 ? launch<true>(device, hp, n_boards, lds_head, logits, st) : launch<false>(device, hp, n_boards, lds_head, logits, st);
}
