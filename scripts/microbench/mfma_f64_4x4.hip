// Micro-benchmark: v_mfma_f64_4x4x4_4b_f64 (four independent 4×4×4 blocks per instruction) on gfx950 — issue rate against the 16×16×4 form,
// and the lane layout of its operands, found by trying the candidate maps against a host product.
// Measured (MI355X): 4x4x4_4b 74.6 TFLOP/s = 16.9 cycles per instruction per SIMD; 16x16x4 47–49 TFLOP/s = 102–107 cycles (64 nominal).
// Layout: A: lane = i + 4·block + 16·k;  B: lane = j + 4·block + 16·k;  D: lane = j + 4·block + 16·i.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_f64_4x4.hip -o mfma_f64_4x4.bin ; run: ./mfma_f64_4x4.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_4x4(double *out, int iters, double a0, double b0)
{
    double acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = 0.0;
    double a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the same with NC dependent chains per wave (each product waits for the previous one of its chain)
template <int NC>
__global__ void __launch_bounds__(256) k_4x4_chains(double *out, int iters, double a0, double b0)
{
    double acc[NC];
    for (int i = 0; i < NC; ++i) acc[i] = 0.0;
    double a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16 / NC; ++r)
#pragma unroll
            for (int i = 0; i < NC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_16(double *out, int iters, double a0, double b0)
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

__global__ void k_probe(const double *a, const double *b, double *d)
{
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}

int main()
{
    double *d; hipMalloc(&d, sizeof(double) * 256 * 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8, iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        float ms;
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_4x4, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0, 2.0);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        const double n4 = (double)blocks * 4 * iters * 16;
        printf("mfma_f64_4x4x4_4b: %.3f ms  %.1f TFLOP/s  (%.1f cycles per instruction per SIMD at 2.4 GHz)\n", ms, n4 * 512 / ms * 1e-9, ms * 1e-3 * 2.4e9 / (n4 / 1024));
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_16, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0, 2.0);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        const double n16 = (double)blocks * 4 * iters * 8;
        printf("mfma_f64_16x16x4:  %.3f ms  %.1f TFLOP/s  (%.1f cycles per instruction per SIMD at 2.4 GHz)\n", ms, n16 * 2048 / ms * 1e-9, ms * 1e-3 * 2.4e9 / (n16 / 1024));
    }
    {   // dependent chains, one wave per SIMD (256 threads per CU): issue interval of a chained product
        float ms;
        const int b1 = 256;
        auto run = [&](auto kern, int nc) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(b1), dim3(256), 0, 0, d, iters, 1.0, 2.0);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            printf("4x4x4_4b, %2d chain(s), one wave per SIMD: %.1f cycles per instruction at 2.4 GHz\n", nc, ms * 1e-3 * 2.4e9 / ((double)iters * 16));
        };
        run(k_4x4_chains<1>, 1); run(k_4x4_chains<1>, 1); run(k_4x4_chains<2>, 2); run(k_4x4_chains<4>, 4); run(k_4x4_chains<8>, 8); run(k_4x4_chains<16>, 16);
    }
    // layout: random integers, all candidate maps
    double ha[64], hb[64], hd[64], *da, *db, *dd;
    for (int l = 0; l < 64; ++l) { ha[l] = (l * 7 + 3) % 11 - 5; hb[l] = (l * 5 + 1) % 13 - 6; }
    hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dd, 512);
    hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, da, db, dd);
    hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost);
    // candidates: lane = p0 + 4·p1 + 16·p2; every assignment of (row, block, k) / (col, block, k) / (row, col, block) to (p0, p1, p2)
    int found = 0;
    const int perm[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
    for (int pa = 0; pa < 6; ++pa)
        for (int pb = 0; pb < 6; ++pb)
            for (int pd = 0; pd < 6; ++pd) {
                double A[4][4][4], B[4][4][4];
                for (int l = 0; l < 64; ++l) {
                    const int p[3] = {l & 3, (l >> 2) & 3, l >> 4};
                    A[p[perm[pa][1]]][p[perm[pa][0]]][p[perm[pa][2]]] = ha[l]; // [block][i][k]
                    B[p[perm[pb][1]]][p[perm[pb][2]]][p[perm[pb][0]]] = hb[l]; // [block][k][j]
                }
                int ok = 1;
                for (int l = 0; l < 64 && ok; ++l) {
                    const int p[3] = {l & 3, (l >> 2) & 3, l >> 4};
                    const int i = p[perm[pd][0]], j = p[perm[pd][1]], blk = p[perm[pd][2]];
                    double sum = 0;
                    for (int k = 0; k < 4; ++k) sum += A[blk][i][k] * B[blk][k][j];
                    if (sum != hd[l]) ok = 0;
                }
                if (ok) {
                    found = 1;
                    const char *nm[3] = {"lane & 3", "(lane >> 2) & 3", "lane >> 4"};
                    printf("layout: A[i][k] of block b: i = %s, b = %s, k = %s;  B[k][j]: j = %s, b = %s, k = %s;  D[i][j]: i = %s, j = %s, b = %s\n", nm[perm[pa][0]],
                           nm[perm[pa][1]], nm[perm[pa][2]], nm[perm[pb][0]], nm[perm[pb][1]], nm[perm[pb][2]], nm[perm[pd][0]], nm[perm[pd][1]], nm[perm[pd][2]]);
                }
            }
    if (!found) { printf("no candidate layout matches; D =\n"); for (int l = 0; l < 64; ++l) printf("%g%c", hd[l], l % 16 == 15 ? '\n' : ' '); }
    return 0;
}
