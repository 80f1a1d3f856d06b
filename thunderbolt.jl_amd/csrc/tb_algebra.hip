// tb_algebra.hip — the heat-step algebra around the assembled operators (all HBM-bound streams):
//   Anz = Mnz − Δt·Knz                src/solver/time/euler.jl:110-116
//   y = α·A·x + β·y (CSR)             src/utils.jl:185-231  (`b = M uₙ₋₁`, euler.jl:85)
//   y += a·x                          add!(b, source), euler.jl:90
//   max |x[i·stride]|                 RTC controller input, src/solver/time/rtc.jl:64-73
#include <hip/hip_runtime.h>

#include <cstring>

#include "tb_internal.h"

namespace tb {

static inline unsigned grid_for(tb_device *dev, int64_t n, int bs)
{
    int64_t nb = (n + bs - 1) / bs;
    const int64_t cap = (int64_t)dev->n_cu * 8;
    return (unsigned)(nb > cap ? cap : nb);
}

__global__ void __launch_bounds__(256)
k_heat_matrix(int64_t n, const double *__restrict__ M, const double *__restrict__ K, double dt, double *__restrict__ A)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t n2 = n >> 1;
    const double2 *M2 = (const double2 *)M;
    const double2 *K2 = (const double2 *)K;
    double2 *A2 = (double2 *)A;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
        const double2 m = M2[i], k = K2[i];
        A2[i] = make_double2(m.x - dt * k.x, m.y - dt * k.y);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) A[n - 1] = M[n - 1] - dt * K[n - 1];
}

__global__ void __launch_bounds__(256) k_axpy(int64_t n, double a, const double *__restrict__ x, double *__restrict__ y)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] += a * x[i];
}

// CSR SpMV, LANES lanes per row (FE rows hold ~27 nz): row-contiguous reads of nzval / colidx
template <int LANES>
__global__ void __launch_bounds__(256)
k_spmv(int64_t nrows, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ colidx, const double *__restrict__ nz,
       const double *__restrict__ x, double alpha, double beta, double *__restrict__ y)
{
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int sub = threadIdx.x % LANES;
    const int64_t nsub = ((int64_t)gridDim.x * blockDim.x) / LANES;
    for (int64_t r = gid / LANES; r < nrows; r += nsub) {
        const int64_t k0 = rowptr[r], k1 = rowptr[r + 1];
        double v = 0.0;
        for (int64_t k = k0 + sub; k < k1; k += LANES) v += nz[k] * x[colidx[k]];
#pragma unroll
        for (int o = LANES / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, LANES);
        if (sub == 0) y[r] = beta == 0.0 ? alpha * v : alpha * v + beta * y[r];
    }
}

__global__ void __launch_bounds__(256)
k_absmax(int64_t n, const double *__restrict__ x, int64_t stride_x, unsigned long long *__restrict__ out)
{
    double m = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) m = fmax(m, fabs(x[i * stride_x]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
    __shared__ double sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3]));
        // non-negative doubles order like their bit patterns
        atomicMax(out, (unsigned long long)__double_as_longlong(m));
    }
}

int launch_heat_matrix(tb_device *dev, int64_t nnz, const double *M, const double *K, double dt, double *A)
{
    hipLaunchKernelGGL(k_heat_matrix, dim3(grid_for(dev, (nnz + 1) / 2, 256)), dim3(256), 0, dev->stream, nnz, M, K, dt, A);
    TB_HIP(hipGetLastError());
    return TB_OK;
}

int launch_axpy(tb_device *dev, int64_t n, double a, const double *x, double *y)
{
    hipLaunchKernelGGL(k_axpy, dim3(grid_for(dev, n, 256)), dim3(256), 0, dev->stream, n, a, x, y);
    TB_HIP(hipGetLastError());
    return TB_OK;
}

int launch_spmv(tb_pattern *p, const double *nz, const double *x, double alpha, double beta, double *y)
{
    tb_device *dev = p->mesh->dev;
    constexpr int LANES = 8;
    hipLaunchKernelGGL(k_spmv<LANES>, dim3(grid_for(dev, p->n_rows * LANES, 256)), dim3(256), 0, dev->stream, p->n_rows, p->d_rowptr,
                       p->d_colidx, nz, x, alpha, beta, y);
    TB_HIP(hipGetLastError());
    return TB_OK;
}

int launch_absmax(tb_device *dev, int64_t n, const double *x, int64_t stride, double *result)
{
    unsigned long long *d_out = (unsigned long long *)&dev->d_status->cell; // 8-byte scratch inside the status block
    TB_HIP(hipMemsetAsync(d_out, 0, sizeof(unsigned long long), dev->stream));
    hipLaunchKernelGGL(k_absmax, dim3(grid_for(dev, n, 256)), dim3(256), 0, dev->stream, n, x, stride, d_out);
    TB_HIP(hipGetLastError());
    unsigned long long bits = 0;
    TB_HIP(hipMemcpyAsync(&bits, d_out, sizeof bits, hipMemcpyDeviceToHost, dev->stream));
    TB_HIP(hipStreamSynchronize(dev->stream));
    memcpy(result, &bits, sizeof bits);
    return TB_OK;
}

} // namespace tb
