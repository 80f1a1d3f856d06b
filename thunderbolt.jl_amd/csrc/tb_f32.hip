// tb_f32.hip — Float32 value type of the boundary: `MI355XDevice{Float32,Int32}`.
//
// The reference types its device path by value_type(device) and its own GPU tests run Float32 (ext/CuThunderboltExt.jl:126-127,
// test/gpu/test_operators.jl:20-31, test/gpu/ensemble-test.jl, test/gpu/diffusion-test.jl).  The kernels of this library compute in Float64; the
// *_f32 entry points keep every caller-visible array in Float32 and convert at the boundary: assembly and the Krylov solve run the Float64 kernels on
// a scratch arena owned by the device object and round the result once (so a Float32 operator is the correctly rounded Float64 one — at least as
// accurate as a Float32 evaluation of the same sums); the pointwise reaction step, SpMV, A = M − Δt K and axpy read and write Float32 directly with
// Float64 arithmetic in registers.  Not the fast path (the arena costs one extra pass over the output); the drop-in path for Float32 call sites.
#include <hip/hip_runtime.h>

#include "tb_internal.h"

namespace tb {

static inline unsigned f32_grid(tb_device *dev, int64_t n)
{
    const int64_t nb = (n + 255) / 256, cap = (int64_t)dev->n_cu * 32;
    return (unsigned)(nb < 1 ? 1 : nb > cap ? cap : nb);
}

__global__ void __launch_bounds__(256) k_f64_to_f32(int64_t n, const double *__restrict__ in, float *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (float)in[i];
}
__global__ void __launch_bounds__(256) k_f32_to_f64(int64_t n, const float *__restrict__ in, double *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (double)in[i];
}
__global__ void __launch_bounds__(256) k_heat_matrix_f32(int64_t n, const float *__restrict__ M, const float *__restrict__ K, double dt, float *__restrict__ A)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) A[i] = (float)((double)M[i] - dt * (double)K[i]);
}
__global__ void __launch_bounds__(256) k_axpy_f32(int64_t n, double a, const float *__restrict__ x, float *__restrict__ y)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = (float)((double)y[i] + a * (double)x[i]);
}
// y = α A x + β y, 16 lanes per row, Float32 storage, Float64 accumulation (src/utils.jl:185-231)
__global__ void __launch_bounds__(256)
k_spmv_f32(int64_t nrows, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ colidx, const float *__restrict__ nz, const float *__restrict__ x,
           double alpha, double beta, float *__restrict__ y)
{
    constexpr int LN = 16;
    const int sub = threadIdx.x % LN;
    const int64_t nsub = ((int64_t)gridDim.x * blockDim.x) / LN;
    for (int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / LN; r < nrows; r += nsub) {
        double v = 0.0;
        for (int64_t k = rowptr[r] + sub; k < rowptr[r + 1]; k += LN) v += (double)nz[k] * (double)x[colidx[k]];
#pragma unroll
        for (int o = LN / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, LN);
        if (sub == 0) y[r] = (float)(alpha * v + (beta != 0.0 ? beta * (double)y[r] : 0.0));
    }
}

// scratch arena of the device object (grown on demand, released with the device)
static int arena(tb_device *dev, size_t bytes, double **out)
{
    TB_HIP(hipSetDevice(dev->id)); // every *_f32 entry comes through here first: the arena (and the conversion kernels behind it) belong to this device, whatever the caller's current one
    if (dev->scratch_bytes < bytes) {
        TB_SYNC_STREAM(dev);
        if (dev->d_scratch) (void)hipFree(dev->d_scratch);
        dev->d_scratch = nullptr; dev->scratch_bytes = 0;
        hipError_t e = hipMalloc((void **)&dev->d_scratch, bytes);
        if (e != hipSuccess) { set_error("Float32 boundary: scratch arena of %zu B: %s", bytes, hipGetErrorString(e)); return TB_ERR_NOMEM; }
        dev->scratch_bytes = bytes;
    }
    *out = (double *)dev->d_scratch;
    return TB_OK;
}

static int down(tb_device *dev, int64_t n, const double *in, float *out)
{
    if (n > 0) hipLaunchKernelGGL(k_f64_to_f32, dim3(f32_grid(dev, n)), dim3(256), 0, dev->stream, n, in, out);
    TB_HIP(hipGetLastError());
    return TB_OK;
}
static int up(tb_device *dev, int64_t n, const float *in, double *out)
{
    if (n > 0) hipLaunchKernelGGL(k_f32_to_f64, dim3(f32_grid(dev, n)), dim3(256), 0, dev->stream, n, in, out);
    TB_HIP(hipGetLastError());
    return TB_OK;
}

} // namespace tb

using namespace tb;

extern "C" {

int tb_convert_f64_to_f32(tb_device *dev, int64_t n, const double *d_in, float *d_out)
{
    TB_REQUIRE(dev && n >= 0 && ((d_in && d_out) || n == 0), "tb_convert_f64_to_f32: bad argument");
    TB_HIP(hipSetDevice(dev->id));
    return down(dev, n, d_in, d_out);
}

int tb_convert_f32_to_f64(tb_device *dev, int64_t n, const float *d_in, double *d_out)
{
    TB_REQUIRE(dev && n >= 0 && ((d_in && d_out) || n == 0), "tb_convert_f32_to_f64: bad argument");
    TB_HIP(hipSetDevice(dev->id));
    return up(dev, n, d_in, d_out);
}

int tb_assemble_matrix_f32(tb_form *form, tb_pattern *pat, int strategy, double t, float *d_nzval)
{
    TB_REQUIRE(form && pat && (d_nzval || pat->nnz == 0), "tb_assemble_matrix_f32: NULL argument");
    tb_device *dev = form->mesh->dev;
    double *s = nullptr;
    int rc = arena(dev, sizeof(double) * (size_t)pat->nnz, &s);
    if (rc) return rc;
    rc = tb_assemble_matrix(form, pat, strategy, t, s);
    if (rc) return rc;
    return down(dev, pat->nnz, s, d_nzval);
}

int tb_assemble_matrix_pair_f32(tb_form *mass, tb_form *diffusion, tb_pattern *pat, int strategy, double t, float *d_nzval_mass, float *d_nzval_diffusion)
{
    TB_REQUIRE(mass && diffusion && pat && ((d_nzval_mass && d_nzval_diffusion) || pat->nnz == 0), "tb_assemble_matrix_pair_f32: NULL argument");
    tb_device *dev = mass->mesh->dev;
    double *s = nullptr;
    int rc = arena(dev, 2 * sizeof(double) * (size_t)pat->nnz, &s);
    if (rc) return rc;
    rc = tb_assemble_matrix_pair(mass, diffusion, pat, strategy, t, s, s + pat->nnz);
    if (rc) return rc;
    rc = down(dev, pat->nnz, s, d_nzval_mass);
    if (rc) return rc;
    return down(dev, pat->nnz, s + pat->nnz, d_nzval_diffusion);
}

int tb_assemble_vector_f32(tb_form *form, int strategy, double t, float *d_b)
{
    TB_REQUIRE(form && d_b, "tb_assemble_vector_f32: NULL argument");
    tb_device *dev = form->mesh->dev;
    double *s = nullptr;
    int rc = arena(dev, sizeof(double) * (size_t)form->mesh->ndofs, &s);
    if (rc) return rc;
    rc = tb_assemble_vector(form, strategy, t, s);
    if (rc) return rc;
    return down(dev, form->mesh->ndofs, s, d_b);
}

int tb_reaction_step_f32(tb_device *dev, int model, const double *params, int n_params, float *d_u, float *d_du, int64_t n_points, int n_states, int layout,
                         const float *d_x, int sdim, double t, double dt, int substeps, double threshold)
{
    // one pass: the reaction kernels instantiated on Float32 storage (states read, stepped in Float64, rounded once) — no conversion round trip
    TB_REQUIRE(dev && params && (d_u || n_points == 0) && n_points >= 0 && n_states > 0, "tb_reaction_step_f32: bad argument");
    TB_REQUIRE(model != TB_CELL_FHN_HETEROGENEOUS || d_x || n_points == 0, "tb_reaction_step_f32: this cell model reads the point coordinate — pass d_x");
    TB_REQUIRE(!d_x || (sdim >= 1 && sdim <= 3), "tb_reaction_step_f32: sdim must be 1, 2 or 3 (got %d)", sdim);
    int ns, np;
    int rc = tb_cell_model_info(model, &ns, &np, nullptr);
    if (rc) return rc;
    TB_REQUIRE(n_states == ns, "tb_reaction_step_f32: model has %d states, caller says %d", ns, n_states);
    TB_REQUIRE(n_params == np, "tb_reaction_step_f32: model has %d parameters, caller passed %d", np, n_params);
    TB_REQUIRE(layout == TB_LAYOUT_SOA || layout == TB_LAYOUT_AOS, "tb_reaction_step_f32: unknown layout %d", layout);
    if (n_points == 0) return TB_OK;
    TB_HIP(hipSetDevice(dev->id));
    return launch_reaction_f32(dev, model, params, n_params, d_u, d_du, n_points, layout, t, dt, substeps, threshold, d_x, sdim);
}

int tb_spmv_csr_f32(tb_pattern *pat, const float *d_nzval, const float *d_x, double alpha, double beta, float *d_y)
{
    TB_REQUIRE(pat && d_nzval && d_x && d_y && d_x != d_y, "tb_spmv_csr_f32: bad argument");
    tb_device *dev = pat->mesh->dev;
    TB_HIP(hipSetDevice(dev->id));
    if (pat->n_rows > 0)
        hipLaunchKernelGGL(k_spmv_f32, dim3(f32_grid(dev, pat->n_rows * 16)), dim3(256), 0, dev->stream, pat->n_rows, pat->d_rowptr, pat->d_colidx, d_nzval, d_x, alpha, beta, d_y);
    TB_HIP(hipGetLastError());
    return TB_OK;
}

int tb_heat_matrix_f32(tb_device *dev, int64_t nnz, const float *d_Mnz, const float *d_Knz, double dt, float *d_Anz)
{
    TB_REQUIRE(dev && nnz >= 0 && ((d_Mnz && d_Knz && d_Anz) || nnz == 0), "tb_heat_matrix_f32: bad argument");
    TB_HIP(hipSetDevice(dev->id));
    if (nnz > 0) hipLaunchKernelGGL(k_heat_matrix_f32, dim3(f32_grid(dev, nnz)), dim3(256), 0, dev->stream, nnz, d_Mnz, d_Knz, dt, d_Anz);
    TB_HIP(hipGetLastError());
    return TB_OK;
}

int tb_axpy_f32(tb_device *dev, int64_t n, double a, const float *d_x, float *d_y)
{
    TB_REQUIRE(dev && n >= 0 && ((d_x && d_y) || n == 0), "tb_axpy_f32: bad argument");
    TB_HIP(hipSetDevice(dev->id));
    if (n > 0) hipLaunchKernelGGL(k_axpy_f32, dim3(f32_grid(dev, n)), dim3(256), 0, dev->stream, n, a, d_x, d_y);
    TB_HIP(hipGetLastError());
    return TB_OK;
}

int tb_cg_solve_f32(tb_pattern *pat, const float *d_Anz, const float *d_b, float *d_x, double rtol, double atol, int maxiter, int jacobi, int *iters, double *resnorm)
{
    TB_REQUIRE(pat && d_Anz && d_b && d_x, "tb_cg_solve_f32: NULL argument");
    tb_device *dev = pat->mesh->dev;
    const int64_t n = pat->n_rows, nnz = pat->nnz;
    double *s = nullptr;
    int rc = arena(dev, sizeof(double) * (size_t)(nnz + 2 * n), &s);
    if (rc) return rc;
    if ((rc = up(dev, nnz, d_Anz, s))) return rc;
    if ((rc = up(dev, n, d_b, s + nnz))) return rc;
    if ((rc = up(dev, n, d_x, s + nnz + n))) return rc;
    rc = tb_cg_solve(pat, s, s + nnz, s + nnz + n, rtol, atol, maxiter, jacobi == TB_JACOBI_REUSE ? 1 : jacobi, iters, resnorm); // the arena is rewritten per call: no D⁻¹ reuse
    if (rc) return rc;
    return down(dev, n, s + nnz + n, d_x);
}

} // extern "C"
