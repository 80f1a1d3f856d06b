// Microbenchmark: throughput of LDS FP64 accumulation primitives on gfx950 (cycles per wave-instruction).
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics lds_atomic.hip -o lds_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) k(const int *idx, double *out, int iters, long long *cyc)
{
    extern __shared__ double acc[];
    for (int i = threadIdx.x; i < 6912; i += blockDim.x) acc[i] = 0.0;
    __syncthreads();
    int a[8];
    for (int j = 0; j < 8; ++j) a[j] = idx[(blockIdx.x * 8 + j) * blockDim.x + threadIdx.x];
    double v = 1.0 + threadIdx.x;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MODE == 0) unsafeAtomicAdd(&acc[a[j]], v);
            else if (MODE == 1) { acc[a[j]] += v; }
            else if (MODE == 2) atomicAdd((float *)&acc[a[j]], (float)v);
            else if (MODE == 3) atomicAdd((unsigned long long *)&acc[a[j]], (unsigned long long)it);
            else if (MODE == 4) acc[a[j]] = v;
        }
    }
    long long t1 = clock64();
    __syncthreads();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[threadIdx.x];
}

int main()
{
    const int T = 256, NB = 512, iters = 200;
    std::vector<int> h(NB * 8 * T);
    const char *pat[] = {"unit stride", "stride 27", "random rows*27+pos", "same row (8-way pos)"};
    const char *mode[] = {"ds_add_f64", "read+add+write b64", "ds_add_f32", "ds_add_u64", "ds_write_b64"};
    int *d_idx; double *d_out; long long *d_cyc;
    CK(hipMalloc(&d_idx, h.size() * 4)); CK(hipMalloc(&d_out, NB * T * 8)); CK(hipMalloc(&d_cyc, NB * 8));
    for (int p = 0; p < 4; ++p) {
        srand(1);
        for (int b = 0; b < NB; ++b)
            for (int j = 0; j < 8; ++j)
                for (int t = 0; t < T; ++t) {
                    int v;
                    if (p == 0) v = (t + 256 * j) % 6912;
                    else if (p == 1) v = (t * 27 + j) % 6912;
                    else if (p == 2) v = (rand() % 256) * 27 + (rand() % 27);
                    else v = ((t / 8) * 27 + (t % 8) + j) % 6912;
                    h[(b * 8 + j) * T + t] = v;
                }
        CK(hipMemcpy(d_idx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        for (int m = 0; m < 5; ++m) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            auto launch = [&]() {
                switch (m) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(NB), dim3(T), 6912 * 8, 0, d_idx, d_out, iters, d_cyc); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(NB), dim3(T), 6912 * 8, 0, d_idx, d_out, iters, d_cyc); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(NB), dim3(T), 6912 * 8, 0, d_idx, d_out, iters, d_cyc); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(NB), dim3(T), 6912 * 8, 0, d_idx, d_out, iters, d_cyc); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(NB), dim3(T), 6912 * 8, 0, d_idx, d_out, iters, d_cyc); break;
                }
            };
            launch(); CK(hipDeviceSynchronize());
            hipEventRecord(e0); launch(); hipEventRecord(e1); CK(hipEventSynchronize(e1));
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<long long> c(NB); CK(hipMemcpy(c.data(), d_cyc, NB * 8, hipMemcpyDeviceToHost));
            double avg = 0; for (auto x : c) avg += x; avg /= NB;
            // per-CU: NB/256 blocks sequentially (2 resident); wave-instr per block = 4 waves * iters * 8
            printf("%-24s %-22s  %.3f ms  block cycles %.0f  -> %.1f cycles per wave-instr per CU (2 WG/CU)\n", pat[p], mode[m], ms, avg,
                   ms * 1e-3 * 2.4e9 / ((double)NB / 256 * 4 * iters * 8));
        }
    }
    return 0;
}
