// tb_assembly.hip — gfx950 element-integration kernels: mass / diffusion matrices scattered into CSR and
// the analytical-source vector, in four scatter strategies (atomic, per-colour, element-assembly, patch).
//
// Arithmetic restated from (paths relative to the Thunderbolt.jl v0.0.4 tree):
//   src/modeling/core/mass.jl:28-43, src/modeling/core/diffusion.jl:28-50 (+ src/utils.jl:409-410),
//   src/modeling/core/analytical_coefficient.jl:80-101, src/modeling/core/coefficients.jl:85-99,152-162,279-292,
//   src/modeling/microstructure.jl:136-138,176-187, src/utils.jl:131-139.
// The cell loop / scatter replaced here is FerriteOperators' (third party); its in-tree model is
// src/modeling/core/coordinate_systems.jl:145-171.
//
// Design notes:
//  * one THREAD per cell, Kₑ kept symmetric-packed in registers; all reference-element values are
//    compile-time immediates (tb_elem.hpp), so the only global reads are connectivity, coordinates and
//    the scatter metadata.  FP64 MFMA runs at the FP64 vector rate on CDNA4, so the Kₑ contraction stays
//    on the VALU.
//  * PATCH strategy: a workgroup owns a Morton-compact patch of cells and the rows first touched by it;
//    it recomputes the halo cells, accumulates every owned row in LDS (ds_add_f64) and writes each
//    nz / dof exactly once with row-contiguous stores — no global atomics, no zero-fill pass.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "tb_elem.hpp"
#include "tb_internal.h"

namespace tb {
using namespace tbk;

struct MeshView {
    const double *xyz;
    const int32_t *conn;
    const int32_t *cell_dofs;
    int64_t n_cells;
};

struct FormArgs {
    double D[9];      // constant tensor (already divided by Cₘ·χ when wrapped)
    double rho;       // constant density
    double lambda[3]; // eigenvalues for the spectral field coefficient
    double scale;     // 1/(Cₘ·χ) for field coefficients (1 when not wrapped)
    const double *field;
    // source
    int src_kind;
    double p0;
    const double *table;
    double t;
};

template <int NB> __host__ __device__ constexpr int sym_idx(int i, int j)
{
    return i <= j ? i * NB - (i * (i - 1)) / 2 + (j - i) : j * NB - (j * (j - 1)) / 2 + (i - j);
}

template <class E>
__device__ __forceinline__ void load_coords(const MeshView &m, int64_t cell, double (&x)[E::NV][3])
{
    int32_t nodes[E::NV];
    const int32_t *c = m.conn + cell * E::NV;
#pragma unroll
    for (int a = 0; a < E::NV; ++a) nodes[a] = c[a];
#pragma unroll
    for (int a = 0; a < E::NV; ++a) {
        const double *p = m.xyz + 3 * (int64_t)nodes[a];
        x[a][0] = p[0]; x[a][1] = p[1]; x[a][2] = p[2];
    }
}

// orthogonalize_system(f,s,n): normalise, then Gram–Schmidt (src/utils.jl:131-139)
__device__ __forceinline__ void orthonormal_frame(double (&f)[3], double (&s)[3], double (&n)[3])
{
    const double rf = 1.0 / sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
    const double rs = 1.0 / sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
    const double rn = 1.0 / sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
#pragma unroll
    for (int d = 0; d < 3; ++d) { f[d] *= rf; s[d] *= rs; n[d] *= rn; }
    const double fs = f[0] * s[0] + f[1] * s[1] + f[2] * s[2];
#pragma unroll
    for (int d = 0; d < 3; ++d) s[d] -= fs * f[d];
    const double fn = f[0] * n[0] + f[1] * n[1] + f[2] * n[2];
    const double sn = s[0] * n[0] + s[1] * n[1] + s[2] * n[2];
#pragma unroll
    for (int d = 0; d < 3; ++d) n[d] = n[d] - fn * f[d] - sn * s[d];
}

// Kₑ / Mₑ of one cell.  SYM: upper triangle packed (requires symmetric D), else full row-major.
template <class E, int FORM, bool FIELD, bool SYM>
__device__ __forceinline__ bool element_matrix(const double (&x)[E::NV][3], const FormArgs &fa, int64_t cell,
                                               double (&Ke)[SYM ? E::NB *(E::NB + 1) / 2 : E::NB * E::NB])
{
    constexpr int NB = E::NB;
    bool ok = true;
    for_each_qp<0, E::NQ>([&](auto qc) {
        constexpr int Q = decltype(qc)::value;
        Geom g;
        geometry<E, Q>(x, g);
        ok = ok && (g.dOmega > 0.0);
        if constexpr (FORM == TB_FORM_MASS) {
            double r = fa.rho;
            if constexpr (FIELD) {
                r = 0.0;
#pragma unroll
                for (int a = 0; a < NB; ++a) r += E::N(Q, a) * fa.field[cell * NB + a];
            }
            const double rw = r * g.dOmega;
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int j = SYM ? i : 0; j < NB; ++j)
                    Ke[SYM ? sym_idx<NB>(i, j) : i * NB + j] += rw * (E::N(Q, i) * E::N(Q, j));
        } else {
            double grad[NB][3];
            mapped_gradients<E, Q>(g, grad);
            double D[3][3];
            if constexpr (FIELD) {
                double f[3] = {0, 0, 0}, s[3] = {0, 0, 0}, n[3] = {0, 0, 0};
                const double *fc = fa.field + cell * (NB * 9);
#pragma unroll
                for (int a = 0; a < NB; ++a) {
                    const double Na = E::N(Q, a);
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        f[d] += Na * fc[9 * a + d];
                        s[d] += Na * fc[9 * a + 3 + d];
                        n[d] += Na * fc[9 * a + 6 + d];
                    }
                }
                orthonormal_frame(f, s, n);
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        D[i][j] = (fa.lambda[0] * f[i] * f[j] + fa.lambda[1] * s[i] * s[j] + fa.lambda[2] * n[i] * n[j]) * fa.scale;
            } else {
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) D[i][j] = fa.D[3 * i + j];
            }
            const double mw = -g.dOmega;
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                // T = −dΩ · D · ∇Nᵢ ;  Kₑ[i,j] += ∇Nⱼ · T   ( = −(∇Nⱼ·D·∇Nᵢ) dΩ, diffusion.jl:44 )
                double T[3];
#pragma unroll
                for (int m = 0; m < 3; ++m) T[m] = mw * (D[m][0] * grad[i][0] + D[m][1] * grad[i][1] + D[m][2] * grad[i][2]);
#pragma unroll
                for (int j = SYM ? i : 0; j < NB; ++j)
                    Ke[SYM ? sym_idx<NB>(i, j) : i * NB + j] += grad[j][0] * T[0] + grad[j][1] * T[1] + grad[j][2] * T[2];
            }
        }
    });
    return ok;
}

__device__ __forceinline__ double eval_source(const FormArgs &fa, const double (&xq)[3], int64_t cell, int q, int nq)
{
    switch (fa.src_kind) {
    case TB_SRC_CONST: return fa.p0;
    case TB_SRC_NORM_PLUS_T: return sqrt(xq[0] * xq[0] + xq[1] * xq[1] + xq[2] * xq[2]) + fa.t;
    case TB_SRC_COS_EXP: {
        const double nr = sqrt(xq[0] * xq[0] + xq[1] * xq[1] + xq[2] * xq[2]);
        return cos(2.0 * 3.141592653589793 * fa.t) * exp(-(nr * nr));
    }
    case TB_SRC_TABULATED: return fa.table[cell * nq + q];
    }
    return 0.0;
}

// bₑ[j] += f(x_q,t)·Nⱼ·dΩ  (analytical_coefficient.jl:89-99)
template <class E>
__device__ __forceinline__ bool element_source(const double (&x)[E::NV][3], const FormArgs &fa, int64_t cell, double (&be)[E::NB])
{
    bool ok = true;
    for_each_qp<0, E::NQ>([&](auto qc) {
        constexpr int Q = decltype(qc)::value;
        Geom g;
        geometry<E, Q>(x, g);
        ok = ok && (g.dOmega > 0.0);
        double xq[3] = {0, 0, 0};
#pragma unroll
        for (int a = 0; a < E::NV; ++a)
#pragma unroll
            for (int d = 0; d < 3; ++d) xq[d] += E::M(Q, a) * x[a][d];
        const double fw = eval_source(fa, xq, cell, Q, E::NQ) * g.dOmega;
#pragma unroll
        for (int j = 0; j < E::NB; ++j) be[j] += fw * E::N(Q, j);
    });
    return ok;
}

__device__ __forceinline__ void flag_neg_detj(Status *st, int64_t cell)
{
    st->neg_detj = 1;
    st->cell = cell;
}

// ------------------------------------------------------------------------------------------------
// scatter-map construction: emap[(i*ndpc+j)*n_cells + cell] = nz index of (dofs[i], dofs[j])
// ------------------------------------------------------------------------------------------------
template <class MapT>
__global__ void k_build_emap(const int32_t *__restrict__ cell_dofs, int64_t n_cells, int ndpc,
                             const int64_t *__restrict__ rowptr, const int32_t *__restrict__ colidx, MapT *__restrict__ emap,
                             Status *st)
{
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= n_cells * ndpc) return;
    // lane ↔ cell (coalesced emap writes), i is the slow index
    const int64_t cell = tid % n_cells;
    const int i = (int)(tid / n_cells);
    const int32_t *d = cell_dofs + cell * ndpc;
    const int32_t row = d[i];
    const int64_t lo0 = rowptr[row], hi0 = rowptr[row + 1];
    for (int j = 0; j < ndpc; ++j) {
        const int32_t c = d[j];
        int64_t lo = lo0, hi = hi0;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (colidx[mid] < c) lo = mid + 1; else hi = mid;
        }
        if (lo >= hi0 || colidx[lo] != c) { st->pattern_missing = 1; st->cell = cell; lo = lo0; }
        emap[(int64_t)(i * ndpc + j) * n_cells + cell] = (MapT)lo;
    }
}

// ------------------------------------------------------------------------------------------------
// direct strategies (atomic / per-colour): one thread per cell
// ------------------------------------------------------------------------------------------------
template <class E, int FORM, bool FIELD, bool SYM, class MapT>
__global__ void __launch_bounds__(128)
k_matrix_direct(MeshView m, FormArgs fa, const MapT *__restrict__ emap, const int32_t *__restrict__ list, int64_t n,
                double *__restrict__ nz, int atomic, Status *st)
{
    constexpr int NB = E::NB;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= n) return;
    const int64_t cell = list ? list[tid] : tid;
    double x[E::NV][3];
    load_coords<E>(m, cell, x);
    double Ke[SYM ? NB * (NB + 1) / 2 : NB * NB];
#pragma unroll
    for (int k = 0; k < (SYM ? NB * (NB + 1) / 2 : NB * NB); ++k) Ke[k] = 0.0;
    if (!element_matrix<E, FORM, FIELD, SYM>(x, fa, cell, Ke)) flag_neg_detj(st, cell);
    const MapT *mp = emap + cell;
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const double v = Ke[SYM ? sym_idx<NB>(i, j) : i * NB + j];
            const int64_t k = (int64_t)mp[(int64_t)(i * NB + j) * m.n_cells];
            if (atomic) unsafeAtomicAdd(nz + k, v); else nz[k] += v;
        }
}

template <class E>
__global__ void __launch_bounds__(256)
k_vector_direct(MeshView m, FormArgs fa, const int32_t *__restrict__ list, int64_t n, double *__restrict__ b,
                double *__restrict__ ea, int mode /*0 atomic, 1 rmw, 2 store to ea*/, Status *st)
{
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= n) return;
    const int64_t cell = list ? list[tid] : tid;
    double x[E::NV][3];
    load_coords<E>(m, cell, x);
    double be[E::NB];
#pragma unroll
    for (int j = 0; j < E::NB; ++j) be[j] = 0.0;
    if (!element_source<E>(x, fa, cell, be)) flag_neg_detj(st, cell);
    const int32_t *d = m.cell_dofs + cell * E::NB;
#pragma unroll
    for (int j = 0; j < E::NB; ++j) {
        if (mode == 2) ea[cell * E::NB + j] = be[j];
        else if (mode == 0) unsafeAtomicAdd(b + d[j], be[j]);
        else b[d[j]] += be[j];
    }
}

// element-assembly gather: b[d] = Σ (in cell order) bₑ slots of dof d
__global__ void k_ea_gather(const int64_t *__restrict__ ptr, const int32_t *__restrict__ src, const double *__restrict__ ea,
                            int64_t ndofs, double *__restrict__ b)
{
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= ndofs) return;
    double s = 0.0;
    for (int64_t k = ptr[d]; k < ptr[d + 1]; ++k) s += ea[src[k]];
    b[d] = s;
}

// ------------------------------------------------------------------------------------------------
// PATCH strategy
// ------------------------------------------------------------------------------------------------
struct PatchView {
    const int64_t *elem_ptr, *row_ptr;
    const int32_t *elem_cell;
    const uint16_t *elem_lrow;
    const int32_t *row_dof;
    // matrix
    const uint32_t *row_off;
    const void *colpos;
    const int64_t *rowptr; // CSR row pointer (global)
};

template <class E, int FORM, bool FIELD, bool SYM, class PosT, int THREADS>
__global__ void __launch_bounds__(THREADS)
k_matrix_patch(MeshView m, FormArgs fa, PatchView pv, double *__restrict__ nz, Status *st)
{
    constexpr int NB = E::NB;
    extern __shared__ double acc[];
    const int64_t p = blockIdx.x;
    const int64_t e0 = pv.elem_ptr[p], e1 = pv.elem_ptr[p + 1];
    const int64_t r0 = pv.row_ptr[p], r1 = pv.row_ptr[p + 1];
    const int nrows = (int)(r1 - r0);
    // total accumulator entries of this patch = offset past the last owned row
    const int64_t last_dof = pv.row_dof[r1 - 1];
    const int nacc = (int)(pv.row_off[r1 - 1] + (uint32_t)(pv.rowptr[last_dof + 1] - pv.rowptr[last_dof]));
    for (int k = threadIdx.x; k < nacc; k += THREADS) acc[k] = 0.0;
    __syncthreads();

    for (int64_t e = e0 + threadIdx.x; e < e1; e += THREADS) {
        const int64_t cell = pv.elem_cell[e];
        double x[E::NV][3];
        load_coords<E>(m, cell, x);
        double Ke[SYM ? NB * (NB + 1) / 2 : NB * NB];
#pragma unroll
        for (int k = 0; k < (SYM ? NB * (NB + 1) / 2 : NB * NB); ++k) Ke[k] = 0.0;
        if (!element_matrix<E, FORM, FIELD, SYM>(x, fa, cell, Ke)) flag_neg_detj(st, cell);
        const uint16_t *lr = pv.elem_lrow + e * NB;
        const PosT *cp = (const PosT *)pv.colpos + e * (NB * NB);
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const uint16_t s = lr[i];
            if (s == 0xFFFF) continue;
            const uint32_t base = pv.row_off[r0 + s];
#pragma unroll
            for (int j = 0; j < NB; ++j)
                unsafeAtomicAdd(&acc[base + cp[i * NB + j]], Ke[SYM ? sym_idx<NB>(i, j) : i * NB + j]);
        }
    }
    __syncthreads();
    // write-out: 32 lanes per row (row-contiguous, coalesced), rows round-robin over half-waves
    const int half = threadIdx.x >> 5, lane = threadIdx.x & 31;
    for (int s = half; s < nrows; s += THREADS / 32) {
        const int64_t dof = pv.row_dof[r0 + s];
        const int64_t g0 = pv.rowptr[dof];
        const int len = (int)(pv.rowptr[dof + 1] - g0);
        const uint32_t base = pv.row_off[r0 + s];
        for (int k = lane; k < len; k += 32) nz[g0 + k] = acc[base + k];
    }
}

template <class E, int THREADS>
__global__ void __launch_bounds__(THREADS)
k_vector_patch(MeshView m, FormArgs fa, PatchView pv, double *__restrict__ b, Status *st)
{
    extern __shared__ double acc[];
    const int64_t p = blockIdx.x;
    const int64_t e0 = pv.elem_ptr[p], e1 = pv.elem_ptr[p + 1];
    const int64_t r0 = pv.row_ptr[p], r1 = pv.row_ptr[p + 1];
    const int nrows = (int)(r1 - r0);
    for (int k = threadIdx.x; k < nrows; k += THREADS) acc[k] = 0.0;
    __syncthreads();
    for (int64_t e = e0 + threadIdx.x; e < e1; e += THREADS) {
        const int64_t cell = pv.elem_cell[e];
        double x[E::NV][3];
        load_coords<E>(m, cell, x);
        double be[E::NB];
#pragma unroll
        for (int j = 0; j < E::NB; ++j) be[j] = 0.0;
        if (!element_source<E>(x, fa, cell, be)) flag_neg_detj(st, cell);
        const uint16_t *lr = pv.elem_lrow + e * E::NB;
#pragma unroll
        for (int j = 0; j < E::NB; ++j)
            if (lr[j] != 0xFFFF) unsafeAtomicAdd(&acc[lr[j]], be[j]);
    }
    __syncthreads();
    for (int s = threadIdx.x; s < nrows; s += THREADS) b[pv.row_dof[r0 + s]] = acc[s];
}

// ------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------
static inline unsigned nblocks(int64_t n, int bs) { return (unsigned)((n + bs - 1) / bs); }

int launch_build_emap(tb_pattern *p)
{
    tb_mesh *m = p->mesh;
    tb_device *dev = m->dev;
    const int64_t n = m->n_cells * m->ndpc;
    p->map64 = p->nnz >= (int64_t)0x7fffffff;
    const size_t bytes = (size_t)m->n_cells * m->ndpc * m->ndpc * (p->map64 ? 8 : 4);
    TB_HIP(hipMalloc(&p->d_emap, bytes));
    int rc = reset_status(dev);
    if (rc) return rc;
    if (p->map64)
        hipLaunchKernelGGL(k_build_emap<int64_t>, dim3(nblocks(n, 256)), dim3(256), 0, dev->stream, m->d_cell_dofs, m->n_cells,
                           m->ndpc, p->d_rowptr, p->d_colidx, (int64_t *)p->d_emap, dev->d_status);
    else
        hipLaunchKernelGGL(k_build_emap<int32_t>, dim3(nblocks(n, 256)), dim3(256), 0, dev->stream, m->d_cell_dofs, m->n_cells,
                           m->ndpc, p->d_rowptr, p->d_colidx, (int32_t *)p->d_emap, dev->d_status);
    TB_HIP(hipGetLastError());
    return check_status(dev);
}

static FormArgs make_args(const tb_form *f, double t)
{
    FormArgs a{};
    for (int i = 0; i < 9; ++i) a.D[i] = f->Dconst[i];
    a.rho = f->coef.p[0];
    a.lambda[0] = f->coef.p[0]; a.lambda[1] = f->coef.p[1]; a.lambda[2] = f->coef.p[2];
    a.scale = f->coef.wrap ? 1.0 / (f->coef.Cm * f->coef.chi) : 1.0;
    a.field = f->d_field;
    a.src_kind = f->coef.kind;
    a.p0 = f->coef.p[0];
    a.table = f->d_table;
    a.t = t;
    return a;
}

static MeshView make_view(const tb_mesh *m) { return MeshView{m->d_xyz, m->d_conn, m->d_cell_dofs, m->n_cells}; }

static PatchView make_patch_view(const tb_mesh *m, const tb_pattern *p)
{
    const PatchPlan *pp = m->patches.get();
    PatchView v{};
    v.elem_ptr = pp->d_elem_ptr; v.row_ptr = pp->d_row_ptr; v.elem_cell = pp->d_elem_cell;
    v.elem_lrow = pp->d_elem_lrow; v.row_dof = pp->d_row_dof;
    if (p) {
        const PatchMatPlan *pm = p->patch_mat.get();
        v.row_off = pm->d_row_off;
        v.colpos = pm->d_colpos8 ? (const void *)pm->d_colpos8 : (const void *)pm->d_colpos16;
        v.rowptr = p->d_rowptr;
    }
    return v;
}

template <class E, int FORM, bool FIELD, bool SYM>
static int run_matrix(tb_form *f, tb_pattern *p, int strategy, double t, double *d_nz)
{
    tb_mesh *m = f->mesh;
    tb_device *dev = m->dev;
    const MeshView mv = make_view(m);
    const FormArgs fa = make_args(f, t);
    if (strategy == TB_STRATEGY_PATCH) {
        if (!m->patches) { int rc = build_patch_plan(m, 0); if (rc) return rc; }
        if (!p->patch_mat) { int rc = build_patch_mat_plan(p); if (rc) return rc; }
        const PatchView pv = make_patch_view(m, p);
        const size_t lds = (size_t)p->patch_mat->max_lds_entries * sizeof(double);
        constexpr int T = 256;
        if (p->patch_mat->d_colpos8) {
            auto k = k_matrix_patch<E, FORM, FIELD, SYM, uint8_t, T>;
            TB_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(k, dim3((unsigned)m->patches->n_patches), dim3(T), lds, dev->stream, mv, fa, pv, d_nz, dev->d_status);
        } else {
            auto k = k_matrix_patch<E, FORM, FIELD, SYM, uint16_t, T>;
            TB_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(k, dim3((unsigned)m->patches->n_patches), dim3(T), lds, dev->stream, mv, fa, pv, d_nz, dev->d_status);
        }
        TB_HIP(hipGetLastError());
        return TB_OK;
    }
    TB_HIP(hipMemsetAsync(d_nz, 0, (size_t)p->nnz * sizeof(double), dev->stream));
    auto go = [&](const int32_t *list, int64_t n, int atomic) -> int {
        if (n == 0) return TB_OK;
        if (p->map64)
            hipLaunchKernelGGL((k_matrix_direct<E, FORM, FIELD, SYM, int64_t>), dim3(nblocks(n, 128)), dim3(128), 0, dev->stream, mv, fa,
                               (const int64_t *)p->d_emap, list, n, d_nz, atomic, dev->d_status);
        else
            hipLaunchKernelGGL((k_matrix_direct<E, FORM, FIELD, SYM, int32_t>), dim3(nblocks(n, 128)), dim3(128), 0, dev->stream, mv, fa,
                               (const int32_t *)p->d_emap, list, n, d_nz, atomic, dev->d_status);
        TB_HIP(hipGetLastError());
        return TB_OK;
    };
    if (strategy == TB_STRATEGY_ATOMIC) return go(nullptr, m->n_cells, 1);
    if (strategy == TB_STRATEGY_PER_COLOR) {
        if (!m->colors) { int rc = build_color_plan(m); if (rc) return rc; }
        for (int c = 0; c < m->colors->ncolors; ++c) {
            int rc = go(m->colors->d_cells + m->colors->offsets[c], m->colors->offsets[c + 1] - m->colors->offsets[c], 0);
            if (rc) return rc;
        }
        return TB_OK;
    }
    set_error("matrix assembly: strategy %d not supported (use ATOMIC, PER_COLOR or PATCH)", strategy);
    return TB_ERR_UNSUPPORTED;
}

template <class E, int FORM>
static int run_matrix_coef(tb_form *f, tb_pattern *p, int strategy, double t, double *d_nz)
{
    if (f->field) return run_matrix<E, FORM, true, true>(f, p, strategy, t, d_nz);
    if (FORM == TB_FORM_DIFFUSION && !f->symmetric) return run_matrix<E, FORM, false, false>(f, p, strategy, t, d_nz);
    return run_matrix<E, FORM, false, true>(f, p, strategy, t, d_nz);
}

template <class E>
static int run_matrix_form(tb_form *f, tb_pattern *p, int strategy, double t, double *d_nz)
{
    if (f->kind == TB_FORM_MASS) return run_matrix_coef<E, TB_FORM_MASS>(f, p, strategy, t, d_nz);
    return run_matrix_coef<E, TB_FORM_DIFFUSION>(f, p, strategy, t, d_nz);
}

int launch_assemble_matrix(tb_form *f, tb_pattern *p, int strategy, double t, double *d_nz)
{
    tb_mesh *m = f->mesh;
    int rc = reset_status(m->dev);
    if (rc) return rc;
    if (m->field_kind == TB_HEX8 && f->qorder == 2) rc = run_matrix_form<Hex8<2>>(f, p, strategy, t, d_nz);
    else if (m->field_kind == TB_TET4 && f->qorder == 2) rc = run_matrix_form<Tet4<2>>(f, p, strategy, t, d_nz);
    else {
        set_error("matrix assembly: field kind %d with quadrature order %d not implemented", m->field_kind, f->qorder);
        return TB_ERR_UNSUPPORTED;
    }
    if (rc) return rc;
    return check_status(m->dev);
}

template <class E>
static int run_vector(tb_form *f, int strategy, double t, double *d_b)
{
    tb_mesh *m = f->mesh;
    tb_device *dev = m->dev;
    const MeshView mv = make_view(m);
    const FormArgs fa = make_args(f, t);
    if (strategy == TB_STRATEGY_PATCH) {
        if (!m->patches) { int rc = build_patch_plan(m, 0); if (rc) return rc; }
        const PatchView pv = make_patch_view(m, nullptr);
        constexpr int T = 256;
        hipLaunchKernelGGL((k_vector_patch<E, T>), dim3((unsigned)m->patches->n_patches), dim3(T),
                           (size_t)m->patches->max_rows * sizeof(double), dev->stream, mv, fa, pv, d_b, dev->d_status);
        TB_HIP(hipGetLastError());
        return TB_OK;
    }
    if (strategy == TB_STRATEGY_ELEMENT) {
        if (!m->ea) { int rc = build_ea_plan(m); if (rc) return rc; }
        hipLaunchKernelGGL((k_vector_direct<E>), dim3(nblocks(m->n_cells, 256)), dim3(256), 0, dev->stream, mv, fa,
                           (const int32_t *)nullptr, m->n_cells, d_b, m->ea->d_ea, 2, dev->d_status);
        TB_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_ea_gather, dim3(nblocks(m->ndofs, 256)), dim3(256), 0, dev->stream, m->ea->d_ptr, m->ea->d_src,
                           m->ea->d_ea, m->ndofs, d_b);
        TB_HIP(hipGetLastError());
        return TB_OK;
    }
    TB_HIP(hipMemsetAsync(d_b, 0, (size_t)m->ndofs * sizeof(double), dev->stream));
    if (strategy == TB_STRATEGY_ATOMIC) {
        hipLaunchKernelGGL((k_vector_direct<E>), dim3(nblocks(m->n_cells, 256)), dim3(256), 0, dev->stream, mv, fa,
                           (const int32_t *)nullptr, m->n_cells, d_b, (double *)nullptr, 0, dev->d_status);
        TB_HIP(hipGetLastError());
        return TB_OK;
    }
    if (strategy == TB_STRATEGY_PER_COLOR) {
        if (!m->colors) { int rc = build_color_plan(m); if (rc) return rc; }
        for (int c = 0; c < m->colors->ncolors; ++c) {
            const int64_t n = m->colors->offsets[c + 1] - m->colors->offsets[c];
            if (!n) continue;
            hipLaunchKernelGGL((k_vector_direct<E>), dim3(nblocks(n, 256)), dim3(256), 0, dev->stream, mv, fa,
                               (const int32_t *)(m->colors->d_cells + m->colors->offsets[c]), n, d_b, (double *)nullptr, 1, dev->d_status);
            TB_HIP(hipGetLastError());
        }
        return TB_OK;
    }
    set_error("vector assembly: unknown strategy %d", strategy);
    return TB_ERR_UNSUPPORTED;
}

int launch_assemble_vector(tb_form *f, int strategy, double t, double *d_b)
{
    tb_mesh *m = f->mesh;
    int rc = reset_status(m->dev);
    if (rc) return rc;
    if (m->ncomp != 1) { set_error("source assembly needs a scalar field"); return TB_ERR_UNSUPPORTED; }
    if (m->field_kind == TB_HEX8 && f->qorder == 2) rc = run_vector<Hex8<2>>(f, strategy, t, d_b);
    else if (m->field_kind == TB_HEX8 && f->qorder == 3) rc = run_vector<Hex8<3>>(f, strategy, t, d_b);
    else if (m->field_kind == TB_TET4 && f->qorder == 2) rc = run_vector<Tet4<2>>(f, strategy, t, d_b);
    else {
        set_error("vector assembly: field kind %d with quadrature order %d not implemented", m->field_kind, f->qorder);
        return TB_ERR_UNSUPPORTED;
    }
    if (rc) return rc;
    return check_status(m->dev);
}

} // namespace tb
