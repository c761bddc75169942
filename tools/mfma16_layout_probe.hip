// Operand / result layout of v_mfma_f32_16x16x4_f32 on gfx950, found by experiment: A = one-hot rows, B = one-hot columns.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma16 tools/mfma16_layout_probe.hip && /tmp/mfma16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float* a_in, const float* b_in, float* d_out)
{
    const int l = threadIdx.x;
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a_in[l], b_in[l], acc, 0, 0, 0);
    for (int r = 0; r < 4; r++) d_out[l * 4 + r] = acc[r];
}

int main()
{
    float *a, *b, *d;
    hipMallocManaged(&a, 64 * 4); hipMallocManaged(&b, 64 * 4); hipMallocManaged(&d, 256 * 4);
    // hypothesis: lane l holds A[i = l % 16][k = l / 16] and B[k = l / 16][j = l % 16]
    // A[i][k] = 1 + i + 100 k,  B[k][j] = (k == 0) ? 1000 * (j + 1) : 0   =>  D[i][j] = (1 + i) * 1000 * (j + 1)
    for (int l = 0; l < 64; l++) { a[l] = 1 + (l % 16) + 100 * (l / 16); b[l] = (l / 16 == 0) ? 1000.f * (l % 16 + 1) : 0.f; }
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, a, b, d);
    hipDeviceSynchronize();
    int ok_a = 1, ok_b = 1;
    for (int l = 0; l < 64; l++)
        for (int r = 0; r < 4; r++) {
            const float v = d[l * 4 + r];
            const int j = l % 16;
            const int i_a = 4 * (l / 16) + r;       // layout (a): row = 4 * (lane / 16) + reg
            const int i_b = (l / 16) + 4 * r;       // layout (b): row = lane / 16 + 4 * reg
            if (v != (1 + i_a) * 1000.f * (j + 1)) ok_a = 0;
            if (v != (1 + i_b) * 1000.f * (j + 1)) ok_b = 0;
        }
    printf("D layout row = 4 * (lane / 16) + reg: %s;  row = lane / 16 + 4 * reg: %s\n", ok_a ? "YES" : "no", ok_b ? "YES" : "no");
    // k check: A[i][k] = (k == 2), B[k][j] = 7 * (k == 2) => D = 7 everywhere if both put k at lane / 16
    for (int l = 0; l < 64; l++) { a[l] = (l / 16 == 2) ? 1.f : 0.f; b[l] = (l / 16 == 2) ? 7.f : 0.f; }
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, a, b, d);
    hipDeviceSynchronize();
    int ok_k = 1;
    for (int e = 0; e < 256; e++) if (d[e] != 7.f) ok_k = 0;
    printf("k = lane / 16 for A and B: %s\n", ok_k ? "YES" : "no");
    return 0;
}
