// Rare-path MakePosDef on the device (one-sided Jacobi eigen-clamp) and the synthetic panel
// generator used by bench.py.
#include "gauss_internal.h"
#include <algorithm>

namespace gauss {

// ------------------------------------------------------------------------------------------
// MakePosDef(m1, min_abs_eig)  (util.cpp:302-318) for the rare case that B11 has an eigenvalue
// below min_abs_eig:   B11 <- V max(Lambda, eps) V^T  =  B11 + sum_{lambda_j < eps} (eps - lambda_j) v_j v_j^T
// Eigenpairs by one-sided (Hestenes) Jacobi on G = B11 V: plane rotations make the columns of G
// mutually orthogonal; at convergence g_j = lambda_j v_j and lambda_j = v_j . g_j (sign kept).
// Rotations of a round act on disjoint column pairs (round-robin schedule), one workgroup each.
// G and V are n x n column-major in d_work; n is even (multiple of NB).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum(double v, double* red, int tid)
{
    red[tid] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(256) void jacobi_init_kernel(const double* __restrict__ A, double* __restrict__ G,
                                                          double* __restrict__ V, int n, int* __restrict__ status)
{
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)n * n) return;
    const int r = (int)(idx % n), c = (int)(idx / n);
    const double a = A[(size_t)r * n + c];      // A is symmetric: row-major == column-major
    G[idx] = a;
    V[idx] = (r == c) ? 1.0 : 0.0;
    if (!isfinite(a)) status[2] = 1;
}

__global__ __launch_bounds__(256) void jacobi_round_kernel(double* __restrict__ G, double* __restrict__ V, int n,
                                                           int round, unsigned int* __restrict__ n_rot,
                                                           const int* __restrict__ status)
{
    __shared__ double red[256];
    if (status[2]) return;
    const int i = blockIdx.x;            // pair index within the round, 0 .. n/2-1
    const int m = n - 1;
    int p, q;
    if (i == 0) { p = m; q = round % m; }
    else { p = (round + i) % m; q = (round - i + m) % m; }
    if (p > q) { const int t = p; p = q; q = t; }
    double* gp = G + (size_t)p * n;
    double* gq = G + (size_t)q * n;
    const int tid = threadIdx.x;
    double a = 0, b = 0, g = 0;
    for (int k = tid; k < n; k += 256) {
        const double x = gp[k], y = gq[k];
        a = fma(x, x, a); b = fma(y, y, b); g = fma(x, y, g);
    }
    a = block_sum(a, red, tid);
    b = block_sum(b, red, tid);
    g = block_sum(g, red, tid);
    if (!(fabs(g) > 1e-15 * sqrt(a * b)) || g == 0.0) return;      // already orthogonal
    const double zeta = (b - a) / (2.0 * g);
    const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
    const double c = 1.0 / sqrt(1.0 + t * t);
    const double s = c * t;
    double* vp = V + (size_t)p * n;
    double* vq = V + (size_t)q * n;
    for (int k = tid; k < n; k += 256) {
        const double x = gp[k], y = gq[k];
        gp[k] = c * x - s * y;
        gq[k] = s * x + c * y;
        const double u = vp[k], w = vq[k];
        vp[k] = c * u - s * w;
        vq[k] = s * u + c * w;
    }
    if (tid == 0) atomicAdd(n_rot, 1u);
}

// lam[j] = v_j . g_j ; delta[j] = max(eps - lam_j, 0)
__global__ __launch_bounds__(256) void jacobi_lambda_kernel(const double* __restrict__ G, const double* __restrict__ V,
                                                            int n, double eps, double* __restrict__ delta,
                                                            int* __restrict__ status)
{
    __shared__ double red[256];
    const int j = blockIdx.x, tid = threadIdx.x;
    double s = 0;
    for (int k = tid; k < n; k += 256) s = fma(V[(size_t)j * n + k], G[(size_t)j * n + k], s);
    s = block_sum(s, red, tid);
    if (tid == 0) {
        delta[j] = (s < eps) ? (eps - s) : 0.0;
        if (!isfinite(s)) status[2] = 1;
    }
}

// A[r][c] += sum_j delta_j V[r][j] V[c][j]   (row-major A, column-major V)
__global__ __launch_bounds__(256) void jacobi_apply_kernel(double* __restrict__ A, const double* __restrict__ V,
                                                           const double* __restrict__ delta, int n)
{
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)n * n) return;
    const int r = (int)(idx / n), c = (int)(idx % n);
    double s = 0.0;
    for (int j = 0; j < n; j++) {
        const double d = delta[j];
        if (d != 0.0) s = fma(d * V[(size_t)j * n + r], V[(size_t)j * n + c], s);
    }
    A[idx] += s;
}

void launch_jacobi_clamp(const Prob* d_probs, int prob, const Prob& hp, double* d_work, bool apply, hipStream_t st)
{
    (void)d_probs; (void)prob;
    const int n = hp.Mld;
    double* G = d_work;
    double* V = G + (size_t)n * n;
    double* delta = V + (size_t)n * n;
    unsigned int* d_rot = reinterpret_cast<unsigned int*>(delta + 2 * (size_t)n);
    const int nb2 = (int)(((size_t)n * n + 255) / 256);
    hipLaunchKernelGGL(jacobi_init_kernel, dim3(nb2), dim3(256), 0, st, hp.A, G, V, n, hp.status);
    for (int sweep = 0; sweep < 30; sweep++) {
        hipMemsetAsync(d_rot, 0, sizeof(unsigned int), st);
        for (int r = 0; r < n - 1; r++)
            hipLaunchKernelGGL(jacobi_round_kernel, dim3(n / 2), dim3(256), 0, st, G, V, n, r, d_rot, hp.status);
        unsigned int h_rot = 0;
        hipMemcpyAsync(&h_rot, d_rot, sizeof(unsigned int), hipMemcpyDeviceToHost, st);
        hipStreamSynchronize(st);
        if (h_rot == 0) break;
    }
    hipLaunchKernelGGL(jacobi_lambda_kernel, dim3(n), dim3(256), 0, st, G, V, n, hp.eps, delta, hp.status);
    if (apply) hipLaunchKernelGGL(jacobi_apply_kernel, dim3(nb2), dim3(256), 0, st, hp.A, V, delta, n);
}

// ------------------------------------------------------------------------------------------
// Synthetic panel generator (bench plumbing).  One thread per sample walks along the SNPs with
// two AR(1) latent haplotypes; allele = latent < thr[snp][population]; genotype = sum of the two.
// Same model as gauss_amd/synth.py (Gaussian-copula AR(1) LD, Balding-Nichols thresholds come
// from the host), with a counter-based hash RNG so the output depends only on (seed, snp, sample).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__device__ __forceinline__ void normal2(uint64_t key, float& n0, float& n1)
{
    const uint64_t h = mix64(key);
    const float u0 = ((float)(uint32_t)(h >> 40) + 0.5f) * (1.0f / 16777216.0f);
    const float u1 = ((float)(uint32_t)((h >> 8) & 0xFFFFFFu) + 0.5f) * (1.0f / 16777216.0f);
    const float r = sqrtf(-2.0f * __logf(u0));
    float sn, cs;
    __sincosf(6.28318530718f * u1, &sn, &cs);
    n0 = r * cs;
    n1 = r * sn;
}

__global__ __launch_bounds__(256) void synth_kernel(uint8_t* __restrict__ out, int n_snp, long long ld,
                                                    const int* __restrict__ pop_off, int n_pop, int n_samples,
                                                    const float* __restrict__ thr, const float* __restrict__ rho,
                                                    uint64_t seed)
{
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= n_samples) return;
    int pop = 0;
    while (pop + 1 < n_pop && n >= pop_off[pop + 1]) pop++;
    float z0, z1;
    normal2(seed * 0x100000001B3ull + (uint64_t)n, z0, z1);
    for (int s = 0; s < n_snp; s++) {
        if (s) {
            const float r = rho[s];
            const float q = sqrtf(fmaxf(0.0f, 1.0f - r * r));
            float e0, e1;
            normal2((seed ^ ((uint64_t)s << 32)) + (uint64_t)n * 0x9E3779B1ull + 7ull, e0, e1);
            z0 = r * z0 + q * e0;
            z1 = r * z1 + q * e1;
        }
        const float t = thr[(size_t)s * n_pop + pop];
        out[(size_t)s * ld + n] = (uint8_t)((z0 < t) + (z1 < t));
    }
}

// One-byte genotype rows -> 2-bit packed rows (GAUSS_GENO_2BIT layout) on the device: the resident form of a
// panel.  One thread per output byte (4 samples); population blocks are 16-byte aligned, zero padded.
__global__ __launch_bounds__(256) void pack2bit_kernel(const uint8_t* __restrict__ in, long long ld_in,
                                                       uint8_t* __restrict__ out, long long ld_out, int n_snp,
                                                       const int* __restrict__ pop_off, const int* __restrict__ blk_off,
                                                       int n_pop)
{
    const long long b = (long long)blockIdx.y * 256 + threadIdx.x;     // byte within the row
    const int row = blockIdx.x;                                        // rows in grid.x: grid.y stops at 65 535
    if (b >= ld_out || row >= n_snp) return;
    int q = 0;
    while (q + 1 < n_pop && b >= blk_off[q + 1]) q++;
    const int s0 = (int)(b - blk_off[q]) * 4;
    const int m = pop_off[q + 1] - pop_off[q];
    const uint8_t* src = in + (size_t)row * ld_in + pop_off[q];
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (s0 + k < m) v |= (uint32_t)(src[s0 + k] & 3u) << (2 * k);
    out[(size_t)row * ld_out + b] = (uint8_t)v;
}

void launch_pack2bit(const uint8_t* d_in, long long ld_in, uint8_t* d_out, long long ld_out, int n_snp,
                     const int* d_pop_off, const int* d_blk_off, int n_pop, hipStream_t s)
{
    hipLaunchKernelGGL(pack2bit_kernel, dim3(n_snp, (unsigned)((ld_out + 255) / 256)), dim3(256), 0, s, d_in, ld_in, d_out,
                       ld_out, n_snp, d_pop_off, d_blk_off, n_pop);
}

void launch_synth(uint8_t* d_out, int n_snp, long long ld, const int* d_pop_off, int n_pop, int n_samples,
                  const float* d_thr, const float* d_rho, uint64_t seed, hipStream_t s)
{
    hipLaunchKernelGGL(synth_kernel, dim3((n_samples + 255) / 256), dim3(256), 0, s, d_out, n_snp, ld, d_pop_off,
                       n_pop, n_samples, d_thr, d_rho, seed);
}

// ------------------------------------------------------------------------------------------
// Host -> HBM copy as a small-footprint kernel.  hipMemcpyAsync from pinned memory does NOT run beside a kernel that
// holds every CU: measured on MI355X (tools/h2d_under_load_probe.py), 256 MB take 4.7 ms with the chip idle (56 GB/s) and
// 40 ms -- the rest of the step -- while the Gram kernel runs, whatever the stream, its priority or the number of
// hardware queues.  Pinned host memory is mapped into the device's address space, so a kernel can read it over PCIe
// itself; one of 256 threads, a handful of registers and no LDS fits into what the Gram kernel's workgroups leave free
// on every CU (k_solve_lite.hip) and moves the same 256 MB in 4.8 ms (55 GB/s) alone AND beside the Gram kernel.
// ------------------------------------------------------------------------------------------
typedef uint32_t h2d_u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void h2d_copy_kernel(const h2d_u32x4* __restrict__ src, h2d_u32x4* __restrict__ dst, size_t n16)
{
    __builtin_amdgcn_s_setprio(3);
    constexpr int UN = 4;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UN - 1) * stride < n16; i += UN * stride) {
        h2d_u32x4 v[UN];
#pragma unroll
        for (int u = 0; u < UN; u++) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < UN; u++) dst[i + u * stride] = v[u];
    }
    for (; i < n16; i += stride) dst[i] = __builtin_nontemporal_load(src + i);
}

// bytes: a multiple of 16; both pointers 16-byte aligned; pinned_src from hipHostMalloc (device-visible)
void launch_h2d_copy(void* d_dst, const void* pinned_src, size_t bytes, hipStream_t s)
{
    if (bytes < 16) return;
    const size_t n16 = bytes / 16;
    const int n_wg = (int)std::min<size_t>(512, (n16 + 255) / 256);
    hipLaunchKernelGGL(h2d_copy_kernel, dim3(n_wg), dim3(256), 0, s, (const h2d_u32x4*)pinned_src, (h2d_u32x4*)d_dst, n16);
}

// Matrix exports: export y of the launch, element e = r * width + c of its compact image <- src[r * pitch + c].  Consecutive
// lanes write consecutive doubles of the pinned mirror (full 64-byte PCIe writes) and read all but contiguously; ~56 GB/s.
__global__ __launch_bounds__(256) void export_rows_kernel(const ExportD* __restrict__ ex)
{
    __builtin_amdgcn_s_setprio(3);
    const ExportD d = ex[blockIdx.y];
    const long long n = (long long)d.rows * d.width;
    const long long stride = (long long)gridDim.x * 256;
    constexpr int UN = 4;
    long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; e + (UN - 1) * stride < n; e += UN * stride) {
        double v[UN];
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const long long q = e + u * stride;
            const int r = (int)(q / d.width), c = (int)(q - (long long)r * d.width);
            v[u] = __builtin_nontemporal_load(d.src + (size_t)r * d.pitch + c);
        }
#pragma unroll
        for (int u = 0; u < UN; u++) d.dst[e + u * stride] = v[u];
    }
    for (; e < n; e += stride) {
        const int r = (int)(e / d.width), c = (int)(e - (long long)r * d.width);
        d.dst[e] = __builtin_nontemporal_load(d.src + (size_t)r * d.pitch + c);
    }
}

void launch_export_rows(const ExportD* d_exports, int n, long long max_elems, hipStream_t s)
{
    if (n <= 0 || max_elems <= 0) return;
    const int gx = (int)std::max<long long>(1, std::min<long long>(std::max(1, 512 / n), (max_elems + 1023) / 1024));
    hipLaunchKernelGGL(export_rows_kernel, dim3(gx, n), dim3(256), 0, s, d_exports);
}

}  // namespace gauss
