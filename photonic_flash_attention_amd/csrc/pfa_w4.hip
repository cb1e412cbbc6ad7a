// pfa_w4.hip -- instantiations of the 4-wave x 64-row forward (fa3_fwd_w4_kernel.h) in a translation unit of their own:
// the kernel wants the 512-register budget and is iterated on separately from the 8-wave kernels in pfa_capi.hip.
#include <hip/hip_runtime.h>

#include "fa3_fwd_w4_kernel.h"

namespace pfa {

// dtype: 0 = bf16, 1 = fp16 (PFA_DTYPE_*); out32: fp32 store.  Returns the kernel's address for hipLaunchKernel.
const void* w4_kernel(int dtype, bool causal, bool out32) {
    if (dtype == 0) {
        if (out32) return causal ? (const void*)&fa3_fwd_w4_kernel<__bf16, true, float> : (const void*)&fa3_fwd_w4_kernel<__bf16, false, float>;
        return causal ? (const void*)&fa3_fwd_w4_kernel<__bf16, true, __bf16> : (const void*)&fa3_fwd_w4_kernel<__bf16, false, __bf16>;
    }
    if (out32) return causal ? (const void*)&fa3_fwd_w4_kernel<_Float16, true, float> : (const void*)&fa3_fwd_w4_kernel<_Float16, false, float>;
    return causal ? (const void*)&fa3_fwd_w4_kernel<_Float16, true, _Float16> : (const void*)&fa3_fwd_w4_kernel<_Float16, false, _Float16>;
}

}  // namespace pfa
