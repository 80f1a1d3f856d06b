// What would a sliced-ELLPACK mirror of the 27-point rows buy the CG product?  Synthetic 217³-node grid (the 216³ bench mesh), values stored per
// slice of 64 consecutive rows as v[k][lane] (27 × 64 doubles contiguous), column = row + offset[k]: every value load is a coalesced 512-byte line
// run and every x load a contiguous segment — no LDS staging, no barrier.  Compared with the byte count of the CSR product (8 B per non-zero + vectors).
// build: hipcc --offload-arch=gfx950 -O3 sell_spmv.hip -o sell_spmv.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Off { int o[27]; };

template <int RPL> // rows per lane
__global__ void __launch_bounds__(256) k_sell(const double *__restrict__ v, const double *__restrict__ x, double *__restrict__ y, long n, Off off)
{
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
#pragma unroll
    for (int rr = 0; rr < RPL; ++rr) {
        const long slice = wave * RPL + rr, row = slice * 64 + lane;
        if (row >= n) return;
        const double *vs = v + slice * (27 * 64) + lane;
        double vv[27], xx[27];
#pragma unroll
        for (int k = 0; k < 27; ++k) vv[k] = __builtin_nontemporal_load(vs + 64 * k);
#pragma unroll
        for (int k = 0; k < 27; ++k) { long c = row + off.o[k]; c = c < 0 ? 0 : c >= n ? n - 1 : c; xx[k] = x[c]; }
        double s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) { s0 += vv[3 * t] * xx[3 * t]; s1 += vv[3 * t + 1] * xx[3 * t + 1]; s2 += vv[3 * t + 2] * xx[3 * t + 2]; }
        y[row] = s0 + (s1 + s2);
    }
}

// the CSR → slice transposition a mirror would pay once per matrix: 64 rows × 27 values through LDS
__global__ void __launch_bounds__(256) k_transpose(const double *__restrict__ csr, double *__restrict__ v, long nslices)
{
    __shared__ double s[4][27 * 64 + 1];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long slice = (long)blockIdx.x * 4 + w;
    if (slice >= nslices) return;
    const double *src = csr + slice * (27 * 64);
    for (int i = lane; i < 27 * 64; i += 64) s[w][i] = src[i];
    __builtin_amdgcn_wave_barrier();
    double *dst = v + slice * (27 * 64);
    for (int k = 0; k < 27; ++k) dst[64 * k + lane] = s[w][27 * lane + k];
}

int main()
{
    const long N1 = 217, n = N1 * N1 * N1, nsl = (n + 63) / 64;
    Off off;
    int q = 0;
    for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) off.o[q++] = (int)(dx + N1 * dy + N1 * N1 * dz);
    double *v, *csr, *x, *y;
    CK(hipMalloc(&v, nsl * 27 * 64 * 8)); CK(hipMalloc(&csr, nsl * 27 * 64 * 8)); CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&y, nsl * 64 * 8));
    CK(hipMemset(v, 0, nsl * 27 * 64 * 8)); CK(hipMemset(csr, 0, nsl * 27 * 64 * 8)); CK(hipMemset(x, 0, n * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double gb = (double)nsl * 64 * (27 * 8 + 16) / 1e9;
    auto time = [&](auto launch, const char *name, double bytes_gb) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
        printf("%-34s %.3f ms  %.2f TB/s\n", name, ms, bytes_gb / ms);
    };
    time([&] { hipLaunchKernelGGL(k_sell<1>, dim3((unsigned)((nsl + 3) / 4)), dim3(256), 0, 0, v, x, y, n, off); }, "sell, 1 row per lane", gb);
    time([&] { hipLaunchKernelGGL(k_sell<2>, dim3((unsigned)((nsl / 2 + 4) / 4)), dim3(256), 0, 0, v, x, y, n, off); }, "sell, 2 rows per lane", gb);
    time([&] { hipLaunchKernelGGL(k_sell<4>, dim3((unsigned)((nsl / 4 + 4) / 4)), dim3(256), 0, 0, v, x, y, n, off); }, "sell, 4 rows per lane", gb);
    time([&] { hipLaunchKernelGGL(k_transpose, dim3((unsigned)((nsl + 3) / 4)), dim3(256), 0, 0, csr, v, nsl); }, "csr -> slices (once per matrix)", (double)nsl * 64 * 27 * 16 / 1e9);
    CK(hipDeviceSynchronize());
    return 0;
}
