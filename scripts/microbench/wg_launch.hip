// Workgroup launch / short-workgroup cost on MI355X: what does a grid of 78 596 workgroups × 256 threads cost when each does (a) nothing,
// (b) zeroes its LDS block, (c) one / two / three dependent global loads, as a function of the dynamic LDS size (occupancy)?
// build: hipcc --offload-arch=gfx950 -O3 wg_launch.hip -o wg_launch.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) k(const int *__restrict__ a, int *__restrict__ out, int lds_doubles, int n)
{
    extern __shared__ double lds[];
    const int tid = threadIdx.x;
    if (MODE >= 1) {
        double2 *z = (double2 *)lds;
        for (int i = tid; i < lds_doubles / 2; i += 256) z[i] = make_double2(0.0, 0.0);
        __syncthreads();
    }
    int v = 0;
    if (MODE >= 2) v = a[((size_t)blockIdx.x * 256 + tid) % n];                 // level 1 (coalesced, HBM)
    if (MODE >= 3) v = a[((size_t)(unsigned)v * 64 + tid) % n];                  // level 2, dependent
    if (MODE >= 4) v = a[((size_t)(unsigned)v * 64 + tid * 3) % n];              // level 3, dependent
    if (MODE >= 1 && lds[tid] == 1.2345) out[0] = v;
    if (MODE >= 2 && v == 123456789) out[1] = v;
}

int main()
{
    const int nwg = 78596, n = 64 << 20;
    int *a, *out;
    CK(hipMalloc(&a, (size_t)n * 4)); CK(hipMalloc(&out, 64));
    std::vector<int> h(n);
    unsigned s = 12345;
    for (int i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (int)(s >> 8); }
    CK(hipMemcpy(a, h.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int ldsz[] = {80936, 53000, 40000, 20000, 1024};
    for (int li = 0; li < 5; ++li) {
        for (int mode = 0; mode <= 4; ++mode) {
            const int lds = ldsz[li];
            auto launch = [&]() {
                switch (mode) {
                case 0: hipFuncSetAttribute((const void *)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); hipLaunchKernelGGL(k<0>, dim3(nwg), dim3(256), lds, 0, a, out, lds / 8, n); break;
                case 1: hipFuncSetAttribute((const void *)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); hipLaunchKernelGGL(k<1>, dim3(nwg), dim3(256), lds, 0, a, out, lds / 8, n); break;
                case 2: hipFuncSetAttribute((const void *)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); hipLaunchKernelGGL(k<2>, dim3(nwg), dim3(256), lds, 0, a, out, lds / 8, n); break;
                case 3: hipFuncSetAttribute((const void *)k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); hipLaunchKernelGGL(k<3>, dim3(nwg), dim3(256), lds, 0, a, out, lds / 8, n); break;
                default: hipFuncSetAttribute((const void *)k<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); hipLaunchKernelGGL(k<4>, dim3(nwg), dim3(256), lds, 0, a, out, lds / 8, n); break;
                }
            };
            launch(); launch();
            CK(hipEventRecord(e0));
            for (int r = 0; r < 10; ++r) launch();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("lds %6d B  mode %d (%s): %.3f ms per launch, %.2f us per workgroup-slot round (%d WGs/CU)\n", lds, mode,
                   mode == 0 ? "empty" : mode == 1 ? "zero LDS" : mode == 2 ? "+1 load" : mode == 3 ? "+2 dependent loads" : "+3 dependent loads", ms / 10,
                   ms / 10 * 1e3 / (nwg / (256.0 * (163840 / lds > 8 ? 8 : 163840 / lds))), 163840 / lds > 8 ? 8 : 163840 / lds);
        }
    }
    return 0;
}
