// Micro-benchmark: FP64 MFMA (v_mfma_f64_16x16x4_f64) vs FP64 vector FMA issue rate on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_f64.hip -o mfma_f64 ; run: ./mfma_f64
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_mfma(double *out, int iters, double a0, double b0)
{
    double4_t acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = double4_t{0, 0, 0, 0};
    double a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_fma(double *out, int iters, double a0, double b0)
{
    double acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = i;
    double a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i] = __builtin_fma(a, acc[i], b);
    }
    double s = 0;
    for (int i = 0; i < 32; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// layout probe: A[i][k] = (i==k_sel), exact integers in B → D rows reveal the lane map
__global__ void k_layout(double *out)
{
    const int l = threadIdx.x;
    const double a = (double)((l & 15) * 10 + (l >> 4));       // A[i=l&15][k=l>>4] = 10 i + k
    const double b = (double)(((l >> 4) == 0) ? ((l & 15) == 3 ? 1.0 : 0.0) : 0.0); // B[k][j] = δ(k,0) δ(j,3)
    double4_t c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];  // expect D[i][3] = 10 i
}

int main()
{
    double *d; hipMalloc(&d, sizeof(double) * 256 * 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8, iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0, 2.0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)blocks * 4 * iters * 8 * 2.0 * 16 * 16 * 4;
        printf("mfma_f64_16x16x4: %.3f ms  %.1f TFLOP/s\n", ms, flop / ms * 1e-9);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0, 2.0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        const double flop2 = (double)blocks * 256 * iters * 32 * 2.0;
        printf("v_fma_f64:        %.3f ms  %.1f TFLOP/s\n", ms, flop2 / ms * 1e-9);
    }
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, d);
    double h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        const int col = l & 15, row = (l >> 4) + 4 * r;
        const double expect = col == 3 ? 10.0 * row : 0.0;
        if (h[l * 4 + r] != expect) { ok = 0; printf("layout mismatch lane %d reg %d: %g vs %g\n", l, r, h[l * 4 + r], expect); }
    }
    printf("layout (col=lane&15,row=(lane>>4)+4*reg; A[l&15][l>>4]; B[l>>4][l&15]): %s\n", ok ? "confirmed" : "WRONG");
    return 0;
}
