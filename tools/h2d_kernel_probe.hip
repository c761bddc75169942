// A host -> device copy done by a small-footprint KERNEL (256 threads, a handful of registers, no LDS) that reads pinned host
// memory over PCIe: does it run beside the Gram kernel, where hipMemcpyAsync does not (tools/h2d_under_load_probe.py)?
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o /tmp/libh2dk.so tools/h2d_kernel_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL>
__global__ __launch_bounds__(256) void h2d_copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16)
{
    __builtin_amdgcn_s_setprio(3);
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < UNROLL; u++) dst[i + u * stride] = v[u];
    }
    for (; i < n16; i += stride) dst[i] = __builtin_nontemporal_load(src + i);
}

extern "C" {
static hipStream_t g_st = nullptr;
// copies bytes (a multiple of 16) from pinned host memory to the device with n_wg workgroups; returns the elapsed ms
float h2dk_copy(const void* host_pinned, void* dev, size_t bytes, int n_wg, int unroll)
{
    if (!g_st) {
        int lo = 0, hi = 0;
        hipDeviceGetStreamPriorityRange(&lo, &hi);
        hipStreamCreateWithPriority(&g_st, hipStreamNonBlocking, hi);
    }
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, g_st);
    if (unroll == 8) hipLaunchKernelGGL(h2d_copy_kernel<8>, dim3(n_wg), dim3(256), 0, g_st, (const u32x4*)host_pinned, (u32x4*)dev, bytes / 16);
    else if (unroll == 4) hipLaunchKernelGGL(h2d_copy_kernel<4>, dim3(n_wg), dim3(256), 0, g_st, (const u32x4*)host_pinned, (u32x4*)dev, bytes / 16);
    else hipLaunchKernelGGL(h2d_copy_kernel<1>, dim3(n_wg), dim3(256), 0, g_st, (const u32x4*)host_pinned, (u32x4*)dev, bytes / 16);
    hipEventRecord(b, g_st);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    return ms;
}
}
