// az_head.hip — instantiations and launcher of the head kernels (az_head.h): fc1 + softmax + tanh.
#include "az_head_params.h"
#include "az_head.h"

template <bool X3> static hipError_t launch(int dv, const HeadParams &hp, int n_boards, int lds_head, float *logits, hipStream_t st) {
    hipError_t s;
    if (hp.n_ot > OTG) { // large action space: partial logits over (board tile x output-tile group x half of K), then softmax
        static bool hg_attr[AZ_MAX_DEVICES] = {false};
        if (dv < 0 || dv >= AZ_MAX_DEVICES || !hg_attr[dv]) {
            s = hipFuncSetAttribute((const void *)az_head_gemm_kernel<X3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (s != hipSuccess) return s;
            if (dv >= 0 && dv < AZ_MAX_DEVICES) hg_attr[dv] = true;
        }
        constexpr int lds_gemm = HG_RING * (HG_BT + HG_OT) * (X3 ? 2 : 1) * 1024; // X3: 4 k-step slots of 32 KiB
        const int n_cg = (hp.n_ot + HG_OT - 1) / HG_OT, bts = (n_boards + 16 * HG_BT - 1) / (16 * HG_BT);
        // (board tiles rounded up to a multiple of 8: the kernel's XCD-aware order; workgroups past the batch leave at once)
        hipLaunchKernelGGL(az_head_gemm_kernel<X3>, dim3((bts + 7) / 8 * 8 * n_cg * HG_KSPLIT), dim3((8 + HG_LOADERS) * 64), lds_gemm, st, hp, logits);
        hipLaunchKernelGGL((az_head_softmax_kernel<X3, HG_KSPLIT>), dim3((n_boards + 3) / 4), dim3(256), 0, st, hp, (const float *)logits);
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
