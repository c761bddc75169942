// Does a small-footprint fp64 kernel make progress BESIDE the Gram kernel?  (round 3, after the CU-mask result.)
// gram_kernel<float> holds 4 workgroups per CU: 4 x 104 VGPRs of 512 per SIMD lane and 4 x 32 KB of the 160 KB LDS, i.e. it
// leaves 96 VGPRs per lane, 32 KB of LDS and four wave slots per SIMD free on every CU at all times.  A workgroup of 256
// threads that fits into that remainder never has to wait for a Gram workgroup to retire.  This probe launches a chain of
// dependent launches of such a kernel (one 64 x 64 x 64 fp64 MFMA product staged through 18 KB of LDS in k slabs of 16, plus
// `serial` barrier-separated dependent LDS steps standing in for a tile factorisation) on its own high-priority stream and
// reports how long the chain takes; tools/coresident_probe.py runs it alone and under the bench job.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libcoprobe.so tools/coresident_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int LDK = 18;

template <int PRIO>
__global__ __launch_bounds__(256, 5) void lite_step(const double* __restrict__ in, double* __restrict__ out, int serial)
{
    if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* TA = sm;                   // [64][LDK]
    double* TB = sm + 64 * LDK;        // [64][LDK]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double* src = in + (size_t)blockIdx.x * 2 * 4096;
    f64x4 acc[4];
    for (int n = 0; n < 4; n++) acc[n] = f64x4{0.0, 0.0, 0.0, 0.0};
    for (int ks = 0; ks < 4; ks++) {
        __syncthreads();
        for (int e = tid; e < 64 * 16; e += 256) {
            const int r = e >> 4, c = e & 15;
            TA[r * LDK + c] = src[r * 64 + 16 * ks + c];
            TB[r * LDK + c] = src[4096 + r * 64 + 16 * ks + c];
        }
        __syncthreads();
        const double* ap = TA + (16 * wave + (lane & 15)) * LDK + (lane >> 4);
        const double* bp = TB + (lane & 15) * LDK + (lane >> 4);
#pragma unroll
        for (int k0 = 0; k0 < 16; k0 += 4) {
            const double a = ap[k0];
#pragma unroll
            for (int n = 0; n < 4; n++) acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bp[n * 16 * LDK + k0], acc[n], 0, 0, 0);
        }
    }
    // dependent steps: every thread's value depends on its neighbour's previous one through LDS
    __syncthreads();
    double v = acc[0][0];
    for (int s = 0; s < serial; s++) {
        sm[tid] = v;
        __syncthreads();
        v = fma(sm[(tid + 1) & 255], 0.5, v * 0.25);
        __syncthreads();
    }
    double* dst = out + (size_t)blockIdx.x * 2 * 4096;
    for (int n = 0; n < 4; n++)
        for (int r = 0; r < 4; r++) {
            const int row = 16 * wave + (lane >> 4) + 4 * r, col = 16 * n + (lane & 15);
            dst[row * 64 + col] = acc[n][r] * 1e-3 + v * 1e-9;
            dst[4096 + row * 64 + col] = acc[n][r] * 1e-3;
        }
}

struct Probe {
    hipStream_t st;
    double* buf[2];
    int max_wg;
    std::vector<hipEvent_t> ev;
};

extern "C" {
void* coprobe_create(int max_wg, int high_priority)
{
    Probe* p = new Probe;
    int lo = 0, hi = 0;
    hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipStreamCreateWithPriority(&p->st, hipStreamNonBlocking, high_priority ? hi : lo);
    p->max_wg = max_wg;
    for (int i = 0; i < 2; i++) {
        hipMalloc(&p->buf[i], (size_t)max_wg * 2 * 4096 * sizeof(double));
        hipMemset(p->buf[i], 0, (size_t)max_wg * 2 * 4096 * sizeof(double));
    }
    hipDeviceSynchronize();
    return p;
}
// queue one chain: n_launch dependent launches of n_wg workgroups; returns an index for coprobe_ms
int coprobe_chain(void* h, int n_launch, int n_wg, int serial, int prio)
{
    Probe* p = (Probe*)h;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, p->st);
    for (int i = 0; i < n_launch; i++)
        if (prio) hipLaunchKernelGGL(lite_step<3>, dim3(n_wg), dim3(256), 2 * 64 * LDK * sizeof(double), p->st, p->buf[i & 1], p->buf[(i + 1) & 1], serial);
        else hipLaunchKernelGGL(lite_step<0>, dim3(n_wg), dim3(256), 2 * 64 * LDK * sizeof(double), p->st, p->buf[i & 1], p->buf[(i + 1) & 1], serial);
    hipEventRecord(b, p->st);
    p->ev.push_back(a); p->ev.push_back(b);
    return (int)p->ev.size() / 2 - 1;
}
float coprobe_ms(void* h, int idx)
{
    Probe* p = (Probe*)h;
    hipEventSynchronize(p->ev[2 * idx + 1]);
    float ms = 0;
    hipEventElapsedTime(&ms, p->ev[2 * idx], p->ev[2 * idx + 1]);
    return ms;
}
int coprobe_vgprs()
{
    hipFuncAttributes at;
    hipFuncGetAttributes(&at, reinterpret_cast<const void*>(lite_step<3>));
    return at.numRegs;
}
}
