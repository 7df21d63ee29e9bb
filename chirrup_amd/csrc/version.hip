// ABI version and build target of libchirrup_amd.so.
#include "../../include/chirrup_amd.h"

extern "C" int chirrup_abi_version(void) { return CHIRRUP_ABI_VERSION; }
extern "C" const char *chirrup_target_arch(void) { return "gfx950"; }

#include <hip/hip_runtime.h>

namespace {
__global__ void clock_probe_kernel(const int iters, unsigned long long *out) {
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float v = (float)threadIdx.x;
    for (int i = 0; i < iters; i++) v = v * 1.0000001f + 0.5f;
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) out[0] = c1 - c0, out[1] = r1 - r0;
    if (v == 12345.678f) out[1] = 0;                 // keeps the loop
}
// Launch-cost probe: workgroups that do nothing but hold `lds` bytes of LDS and optionally idle for `sleep` x ~4 us
// (tools/launch_cost.py: what a kernel node of the decode graph costs beyond the work of its workgroups)
__global__ void noop_kernel(const int sleep, unsigned *sink) {
    extern __shared__ unsigned char lds_[];
    for (int i = 0; i < sleep; i++) __builtin_amdgcn_s_sleep(127);
    if (sleep < 0) sink[0] = lds_[threadIdx.x];     // (never: keeps the LDS allocation referenced)
}
}  // namespace

extern "C" int chirrup_noop_launch(int grid, int block, int lds_bytes, int sleep, void *sink, void *stream) {
    if (grid <= 0 || block <= 0 || block > 1024 || lds_bytes < 0 || lds_bytes > 160 * 1024) return CHIRRUP_E_SHAPE;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(noop_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(noop_kernel, dim3(grid), dim3(block), lds_bytes, static_cast<hipStream_t>(stream), sleep, static_cast<unsigned *>(sink));
    return (int)hipGetLastError();
}

extern "C" int chirrup_clock_probe(int iters, void *out, void *stream) {
    if (iters <= 0) return CHIRRUP_E_SHAPE;
    if (!out) return CHIRRUP_E_NULL;
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), iters, static_cast<unsigned long long *>(out));
    return (int)hipGetLastError();
}
