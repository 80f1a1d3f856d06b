// Micro-benchmark: does a buffer that one kernel writes and the next one reads stay on chip (L2 / Infinity Cache) when it is re-used over and over?
// For sizes S: repeat { write S bytes; read S bytes } on ONE buffer, against the same pair walking through a 16 GB arena (every pass touches new memory).
// Build: hipcc --offload-arch=gfx950 -O3 mall_reuse.hip -o mall_reuse.bin
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_write(double2 *p, size_t n, double v)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_double2(v + i, v);
}
__global__ void k_read(const double2 *p, size_t n, double *out)
{
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const double2 v = p[i]; s += v.x + v.y; }
    if (s == 12345.678) out[0] = s;
}
int main()
{
    const size_t arena = (size_t)16 << 30;
    char *buf; double *out;
    if (hipMalloc(&buf, arena) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&out, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t sizes_mb[] = {16, 32, 64, 96, 128, 192, 256, 384, 512, 1024};
    for (size_t smb : sizes_mb) {
        const size_t bytes = smb << 20, n = bytes / 16;
        const int reps = (int)(((size_t)8 << 30) / bytes);
        for (int mode = 0; mode < 2; ++mode) { // 0: same buffer, 1: walking
            float ms;
            hipEventRecord(e0);
            for (int r = 0; r < reps; ++r) {
                double2 *p = (double2 *)(buf + (mode ? (size_t)r * bytes % (arena - bytes) : 0));
                hipLaunchKernelGGL(k_write, dim3(2048), dim3(256), 0, 0, p, n, (double)r);
                hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, 0, p, n, out);
            }
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            printf("%5zu MB  %s  %7.1f us per write+read pair  %6.2f TB/s (bytes written + read)\n", smb, mode ? "walking " : "re-used ", ms * 1e3 / reps, 2.0 * bytes * reps / ms * 1e-9);
        }
    }
    return 0;
}
