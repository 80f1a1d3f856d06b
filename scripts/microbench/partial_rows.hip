// Partial-line writes on MI355X: what does it cost to fill one array from TWO kernels that each write interleaved 24-byte pieces (the layout a
// "direct emission" of singly-owned 3×3 blocks into scalar CSR rows would produce: a node pair is three 24-byte pieces, and its neighbours in the
// row belong to another cell or to the gather), compared with one kernel that writes everything?  Round 5, config 4: decides whether emitting
// the 47 % single-contribution entries of the Q2 tangent straight into the matrix can beat storing them and gathering.
//   mode 0: one kernel, every double once, coalesced                              (what the gather does today)
//   mode 1: kernel A writes the pieces with even index, kernel B the odd ones     (piece = PIECE doubles; two sources per sector)
//   mode 2: as 1, but the second kernel starts only after the first has finished AND 1 GiB of other traffic has passed (no merge in any cache)
// Reported: milliseconds and GB/s of payload; run under `rocprofv3 --pmc WRITE_SIZE` / `TCC_EA0_RDREQ_sum` for the memory-side bytes.
// build: hipcc --offload-arch=gfx950 -O3 partial_rows.hip -o partial_rows.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int PIECE>
__global__ void __launch_bounds__(256) k_pieces(double *__restrict__ out, long long n, int parity /*-1: all*/, double v)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const long long piece = i / PIECE;
        if (parity < 0 || (piece & 1) == parity) out[i] = v;
    }
}
// variant where a lane owns a whole piece of its parity (no idle lanes): lanes write PIECE doubles at piece index 2·t + parity
template <int PIECE>
__global__ void __launch_bounds__(256) k_pieces_dense(double *__restrict__ out, long long npieces2, int parity, double v)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < npieces2 * PIECE; t += stride) {
        const long long pc = t / PIECE;
        const int e = (int)(t - pc * PIECE);
        out[(2 * pc + parity) * PIECE + e] = v;
    }
}
__global__ void __launch_bounds__(256) k_fill(double *__restrict__ out, long long n, double v)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = v;
}

template <int PIECE> int run(double *buf, double *other, long long n, long long nother)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * 16;
    float ms;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_fill, dim3(grid), dim3(256), 0, 0, buf, n, 1.0);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("piece %2d B  mode 0 (one kernel, all)            %8.3f ms  %7.1f GB/s payload\n", PIECE * 8, ms, n * 8.0 / ms * 1e-6);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_pieces_dense<PIECE>, dim3(grid), dim3(256), 0, 0, buf, n / (2 * PIECE), 0, 2.0);
        hipLaunchKernelGGL(k_pieces_dense<PIECE>, dim3(grid), dim3(256), 0, 0, buf, n / (2 * PIECE), 1, 3.0);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("piece %2d B  mode 1 (two kernels, back to back)  %8.3f ms  %7.1f GB/s payload\n", PIECE * 8, ms, n * 8.0 / ms * 1e-6);
        float ms2 = 0.0f, t;
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_pieces_dense<PIECE>, dim3(grid), dim3(256), 0, 0, buf, n / (2 * PIECE), 0, 4.0);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&t, e0, e1)); ms2 += t;
        hipLaunchKernelGGL(k_fill, dim3(grid), dim3(256), 0, 0, other, nother, 0.5); // flush every cache level with unrelated traffic
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_pieces_dense<PIECE>, dim3(grid), dim3(256), 0, 0, buf, n / (2 * PIECE), 1, 5.0);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&t, e0, e1)); ms2 += t;
        if (rep) printf("piece %2d B  mode 2 (two kernels, caches flushed) %8.3f ms  %7.1f GB/s payload\n", PIECE * 8, ms2, n * 8.0 / ms2 * 1e-6);
    }
    return 0;
}

int main(int argc, char **argv)
{
    const long long n = (argc > 1 ? atoll(argv[1]) : 512LL << 20) / 48 * 48; // doubles (default 4 GiB)
    const long long nother = 160LL << 20;                                      // 1.25 GiB
    double *buf, *other;
    CK(hipMalloc(&buf, n * 8)); CK(hipMalloc(&other, nother * 8));
    if (run<3>(buf, other, n, nother)) return 1;   // 24-byte pieces: one 3×3 block row
    if (run<6>(buf, other, n, nother)) return 1;   // 48 bytes
    if (run<16>(buf, other, n, nother)) return 1;  // 128 bytes: whole lines from each source
    return 0;
}
