// nsr_rt.h -- host-side runtime used by nsr_api.cpp: the HIP runtime, nothing else.
// (tests/emu/ shadows this header to run the same API on a CPU for unit tests.)
#pragma once
#include <hip/hip_runtime.h>

namespace nsr {
inline const char *rt_check_last() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? nullptr : hipGetErrorString(e);
}
inline int rt_current_device() { int d = 0; (void)hipGetDevice(&d); return d; }
inline void rt_record(void *event, void *stream) { (void)hipEventRecord((hipEvent_t)event, (hipStream_t)stream); }
template <typename K>
inline const char *rt_allow_lds(K kernel, int bytes) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    return e == hipSuccess ? nullptr : hipGetErrorString(e);
}
}  // namespace nsr
