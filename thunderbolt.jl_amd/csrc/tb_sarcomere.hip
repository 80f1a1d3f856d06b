// Pointwise sarcomere dynamics on the device: the StandaloneSarcomereModel protocol of the reference
// (src/modeling/solid/contraction.jl:150-163: du = sarcomere_rhs!(u, λ(t), dλdt(t), Ca(t))) advanced by forward Euler, one lane per
// point, the 20 states of a point in registers, states stored point-fastest (SoA) like the cell models.
// HBM-bound: 2·20·8 B state traffic + up to 24 B inputs + 16 B outputs per point and launch.
#include "tb_internal.h"
#include "tb_sarcomere.hpp"

namespace tb {
using namespace tbk;

struct SarcomereInputs {
    const double *stretch, *velocity, *calcium; // per point, or nullptr → the scalar below
    double s_stretch, s_velocity, s_calcium;
};

__global__ void __launch_bounds__(256)
k_sarcomere_rdq20(RDQ20Params p, SarcomereInputs in, double *__restrict__ state, int64_t n, double dt, int substeps, int rate_independent,
                  double *__restrict__ tension, double *__restrict__ stiffness)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double u[RDQ20_NSTATES], du[RDQ20_NSTATES];
#pragma unroll
        for (int k = 0; k < RDQ20_NSTATES; ++k) u[k] = state[(int64_t)k * n + i];
        const double lam = in.stretch ? in.stretch[i] : in.s_stretch;
        const double vel = rate_independent ? 0.0 : (in.velocity ? in.velocity[i] : in.s_velocity);
        const double ca = in.calcium ? in.calcium[i] : in.s_calcium;
        for (int s = 0; s < substeps; ++s) {
            rdq20_rhs(p, u, lam, vel, ca, du);
#pragma unroll
            for (int k = 0; k < RDQ20_NSTATES; ++k) u[k] += dt * du[k];
        }
#pragma unroll
        for (int k = 0; k < RDQ20_NSTATES; ++k) state[(int64_t)k * n + i] = u[k];
        if (tension) tension[i] = rdq20_tension(p, u, lam);
        if (stiffness) stiffness[i] = rdq20_stiffness(p, u, lam);
    }
}

static RDQ20Params rdq20_params(const double *q)
{
    RDQ20Params p;
    p.LA = q[0]; p.LM = q[1]; p.LB = q[2]; p.SL0 = q[3]; p.Q = q[4]; p.Kd0 = q[5]; p.aKd = q[6]; p.mu = q[7]; p.gamma = q[8]; p.Koff = q[9];
    p.Kbasic = q[10]; p.r0 = q[11]; p.alpha = q[12]; p.mu0_fP = q[13]; p.mu1_fP = q[14]; p.a_XB = q[15]; p.eps_v = q[16];
    return p;
}

int launch_sarcomere(tb_device *dev, const double *params, double *d_state, int64_t n, const double *d_stretch, const double *d_velocity,
                     const double *d_calcium, double stretch, double velocity, double calcium, double dt, int substeps, int rate_independent,
                     double *d_tension, double *d_stiffness)
{
    const SarcomereInputs in{d_stretch, d_velocity, d_calcium, stretch, velocity, calcium};
    int64_t nb = (n + 255) / 256;
    const int64_t cap = (int64_t)dev->n_cu * 16;
    if (nb > cap) nb = cap;
    hipLaunchKernelGGL(k_sarcomere_rdq20, dim3((unsigned)nb), dim3(256), 0, dev->stream, rdq20_params(params), in, d_state, n, dt, substeps,
                       rate_independent, d_tension, d_stiffness);
    TB_HIP(hipGetLastError());
    return TB_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
// Local problem of the condensed mechanics, pointwise: backward Euler on the internal state at frozen stretch and calcium,
// (Q − Q_known)/Δt = rhs(Q, λ, 0, Ca) — solve_internal_timestep + the corrector dQ/dλ (materials.jl:1403-1568), rate-free form
// (AsRateIndependent: dλ/dt = 0, materials.jl:1575-1632).
// Sixteen lanes own one point (four points per wave).  Lane c differentiates the right-hand side along the regulatory-unit state c
// (one forward-mode direction per lane — the 16 columns of the Jacobian in one pass); a transpose through LDS hands lane r row r of
// J_RR = I/Δt − ∂rhs_R/∂Q_R, and Gauss–Jordan elimination with partial pivoting runs on rows held in registers, the pivot row
// travelling by cross-lane shuffles.  The cross-bridge block is triangular to the chain (∂rhs_R/∂Q_X = 0): its 4×4 system is
// solved by every lane after the chain's increment is known.  The whole wave stays converged — finished points idle under a mask.
// ------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double grp_shfl(double v, int src) { return __shfl(v, src, 64); }
// rotation within a row of 16 lanes on the data-parallel-primitive path (a VALU move, no LDS round trip): row_ror:N = 0x120 + N
template <int N> __device__ __forceinline__ int row_ror(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x120 + N, 0xF, 0xF, false); }
template <int N> __device__ __forceinline__ double row_ror(double v)
{
    return __hiloint2double(row_ror<N>(__double2hiint(v)), row_ror<N>(__double2loint(v)));
}

__device__ __forceinline__ double rcp_fast(double y) // hardware reciprocal + two Newton steps (≤ 2 ulp) instead of the IEEE division sequence
{
    double r = __builtin_amdgcn_rcp(y);
    r = fma(fma(-y, r, 1.0), r, r);
    r = fma(fma(-y, r, 1.0), r, r);
    return r;
}

// solve the 16×16 system held one row per lane (row[], b) within a 16-lane group; returns x[16] on every lane; false if singular
__device__ __forceinline__ bool group_gauss_jordan(double (&row)[16], double b, int base, int r, double (&x)[16])
{
    bool done = false, ok = true;
    double mypiv = 1.0;
    int piv[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        double cand = done ? -1.0 : fabs(row[k]);
        int who = r;
        // all-reduce (max, lowest lane on ties) over the 16 lanes by rotations: every lane ends with the same (cand, who)
#define TB_ARGMAX_STEP(N) { const double oc = row_ror<N>(cand); const int ow = row_ror<N>(who); if (oc > cand || (oc == cand && ow < who)) { cand = oc; who = ow; } }
        TB_ARGMAX_STEP(8) TB_ARGMAX_STEP(4) TB_ARGMAX_STEP(2) TB_ARGMAX_STEP(1)
#undef TB_ARGMAX_STEP
        if (!(cand > 0.0)) ok = false;
        const int src = base | who;
        const double pk = grp_shfl(row[k], src), pb = grp_shfl(b, src);
        const double m = (r == who) ? 0.0 : row[k] * rcp_fast(pk);
#pragma unroll
        for (int j = k + 1; j < 16; ++j) row[j] -= m * grp_shfl(row[j], src);
        b -= m * pb;
        if (r == who) { done = true; mypiv = row[k]; }
        piv[k] = src;
    }
    const double mine = b * rcp_fast(mypiv);
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = grp_shfl(mine, piv[k]);
    return ok;
}

// 4×4 dense solve with partial pivoting, every lane on its own copy
__device__ __forceinline__ bool solve4(double (&A)[4][4], double (&b)[4])
{
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int i = k + 1; i < 4; ++i)
            if (fabs(A[i][k]) > fabs(A[k][k])) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { const double t = A[k][j]; A[k][j] = A[i][j]; A[i][j] = t; }
                const double t = b[k]; b[k] = b[i]; b[i] = t;
            }
        if (!(fabs(A[k][k]) > 0.0)) ok = false;
        const double inv = rcp_fast(A[k][k]);
#pragma unroll
        for (int i = k + 1; i < 4; ++i) {
            const double m = A[i][k] * inv;
#pragma unroll
            for (int j = k; j < 4; ++j) A[i][j] -= m * A[k][j];
            b[i] -= m * b[k];
        }
    }
#pragma unroll
    for (int i = 3; i >= 0; --i) {
        double s2 = b[i];
#pragma unroll
        for (int j = i + 1; j < 4; ++j) s2 -= A[i][j] * b[j];
        b[i] = s2 * rcp_fast(A[i][i]);
    }
    return ok;
}

// Jacobian of the local problem at state q, distributed over the 16 lanes of a group: row r of J_RR in registers, column r of
// ∂rhs_X/∂Q_R (colX), and rhs(q) itself (f, on every lane).  Lane r derives column r by hand (rdq20_rhs_column — the eight neighbour
// rates and the two cross-bridge exchange rates carry all the nonlinearity), the transpose goes through LDS.
struct LocalJacobian { double row[16], colX[4], kPN, kNP, vel; };
__device__ __forceinline__ void local_jacobian(const RDQ20Params &p, const double (&q)[20], double lam, double vel, double ca, double dt, double (*tr)[17], int r,
                                               LocalJacobian &J, double (&f)[20], RDQ20Lin &L)
{
    rdq20_linearise(p, q, lam, vel, ca, L);
    rdq20_rhs_rows(p, L, q, vel, f);
    double col[20];
    rdq20_rhs_column(p, L, q, r, col);
#pragma unroll
    for (int i = 0; i < 16; ++i) tr[i][r] = col[i];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 0; c < 16; ++c) J.row[c] = (c == r ? 1.0 / dt : 0.0) - tr[r][c];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 4; ++k) J.colX[k] = col[16 + k];
    J.kPN = L.kPN; J.kNP = L.kNP; J.vel = vel;
}

// J y = g: g given on every lane (20 values), y returned on every lane
__device__ __forceinline__ bool local_linear_solve(const RDQ20Params &p, const LocalJacobian &J, double dt, const double (&g)[20], double (&y)[20], int base, int r)
{
    double row[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) row[c] = J.row[c];
    double gr = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) gr = (k == r) ? g[k] : gr;
    double xr[16];
    bool ok = group_gauss_jordan(row, gr, base, r, xr);
    // cross-bridge block: J_XX y_X = g_X − J_XR y_R, J_XR = −∂rhs_X/∂Q_R (lane c holds column c)
    double mine = 0.0;
#pragma unroll
    for (int c = 0; c < 16; ++c) mine = (c == r) ? xr[c] : mine;
    double bx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        double part = J.colX[k] * mine;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
        bx[k] = g[16 + k] + part;
    }
    const double rr = p.r0 + p.alpha * smooth_abs(J.vel, p.eps_v), dP = 1.0 / dt + rr + J.kPN, dN = 1.0 / dt + rr + J.kNP;
    double A[4][4] = {{dP, 0, -J.kNP, 0}, {-J.vel, dP, 0, -J.kNP}, {-J.kPN, 0, dN, 0}, {0, -J.kPN, -J.vel, dN}}; // I/Δt − XB_A (contraction.jl:573-578)
    ok = solve4(A, bx) && ok;
#pragma unroll
    for (int k = 0; k < 16; ++k) y[k] = xr[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) y[16 + k] = bx[k];
    return ok;
}

template <bool SENS>
__global__ void __launch_bounds__(256)
k_sarcomere_implicit(RDQ20Params p, SarcomereInputs in, double *__restrict__ Qs, const double *__restrict__ Qks, int64_t n, double dt, double tol, int max_iters,
                     double tmax, double *__restrict__ dQdl, double *__restrict__ dQdv, double *__restrict__ act, int act_stride, int32_t *__restrict__ status,
                     unsigned long long *__restrict__ n_failed, const int32_t *__restrict__ cells, int nq, int64_t n_work)
{
    const bool rate = in.velocity != nullptr || in.s_velocity != 0.0 || act_stride == 3 || dQdv != nullptr;
    __shared__ double s_tr[4][4][16][17];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, grp = lane >> 4, r = lane & 15, base = lane & 48;
    double (*tr)[17] = s_tr[wv][grp];
    // n: points the state arrays are laid out for (stride); n_work: points solved — all of them, or the quadrature points of `cells` (subdomain)
    for (int64_t w0 = ((int64_t)blockIdx.x * 4 + wv) * 4; w0 < n_work; w0 += (int64_t)gridDim.x * 16) {
        const int64_t wi = w0 + grp;
        const bool valid = wi < n_work;
        const int64_t wc = valid ? wi : n_work - 1;
        const int64_t i = cells ? (int64_t)cells[wc / nq] * nq + wc % nq : wc;
        const int64_t ii = i;
        const double qa = Qs[(int64_t)r * n + ii], qb = r < 4 ? Qs[(int64_t)(16 + r) * n + ii] : 0.0;
        const double ka = Qks[(int64_t)r * n + ii], kb = r < 4 ? Qks[(int64_t)(16 + r) * n + ii] : 0.0;
        double q[20], qk[20];
#pragma unroll
        for (int k = 0; k < 16; ++k) { q[k] = grp_shfl(qa, base | k); qk[k] = grp_shfl(ka, base | k); }
#pragma unroll
        for (int k = 0; k < 4; ++k) { q[16 + k] = grp_shfl(qb, base | k); qk[16 + k] = grp_shfl(kb, base | k); }
        const double lam = in.stretch ? in.stretch[ii] : in.s_stretch;
        const double ca = in.calcium ? in.calcium[ii] : in.s_calcium;
        const double vel = in.velocity ? in.velocity[ii] : in.s_velocity; // dλ/dt of the rate-coupled form; 0 in the rate-free one
        bool active = true;
        int code = LOCAL_SUCCESS;
        for (int it = 1; it <= max_iters; ++it) {
            if (!__any(active)) break;
            double f[20], R[20], y[20];
            LocalJacobian J;
            RDQ20Lin L;
            local_jacobian(p, q, lam, vel, ca, dt, tr, r, J, f, L);
            double rn = 0.0;
#pragma unroll
            for (int k = 0; k < 20; ++k) { R[k] = (q[k] - qk[k]) / dt - f[k]; rn += R[k] * R[k]; }
            rn = sqrt(rn);
            const bool ok = local_linear_solve(p, J, dt, R, y, base, r);
            if (active) {
                if (!ok) { code = LOCAL_LINEAR_SOLVE_FAILED; active = false; }
                else {
#pragma unroll
                    for (int k = 0; k < 20; ++k) q[k] -= y[k];
                    if (rn < tol) active = false;
                    else if (it == max_iters) { code = LOCAL_MAX_ITERS; active = false; }
                    else if (rn != rn) { code = LOCAL_CONVERGENCE_FAILURE; active = false; }
                }
            }
        }
        if (code == LOCAL_SUCCESS) {
            bool inb = true;
#pragma unroll
            for (int k = 0; k < 16; ++k) inb = inb && (q[k] >= 0.0);
            if (!inb) code = LOCAL_INFEASIBLE;
        }
        double ql = 0.0, qv = 0.0;
        if constexpr (SENS) {
            // corrector: dQ/dλ = J⁻¹ ∂rhs/∂λ at the converged state
            double g[20], y[20], f[20];
            LocalJacobian J;
            RDQ20Lin L;
            local_jacobian(p, q, lam, vel, ca, dt, tr, r, J, f, L);
#pragma unroll
            for (int k = 0; k < 20; ++k) g[k] = rdq20_rhs_dlam_row(L, q, k);
            const bool ok = local_linear_solve(p, J, dt, g, y, base, r);
            if (!ok && code == LOCAL_SUCCESS) code = LOCAL_LINEAR_SOLVE_FAILED;
            if (code != LOCAL_SUCCESS) {
#pragma unroll
                for (int k = 0; k < 20; ++k) y[k] = 0.0; // the reference returns a zero sensitivity for a failed point (materials.jl:1596-1597)
            }
            ql = y[17] + y[19];
            if (rate) { // second corrector dQ/d(dλ/dt) (materials.jl:1715-1720): only the cross-bridge rows of ∂rhs/∂v are non-zero, so y_R = 0
                double bx[4];
                rdq20_rhs_dvel(L, q, bx);
                const double rr = p.r0 + p.alpha * smooth_abs(vel, p.eps_v), dP = 1.0 / dt + rr + J.kPN, dN = 1.0 / dt + rr + J.kNP;
                double A[4][4] = {{dP, 0, -J.kNP, 0}, {-vel, dP, 0, -J.kNP}, {-J.kPN, 0, dN, 0}, {0, -J.kPN, -vel, dN}};
                const bool ok2 = solve4(A, bx);
                qv = (ok2 && code == LOCAL_SUCCESS) ? bx[1] + bx[3] : 0.0;
                if (dQdv && valid && r < 4) {
                    double mine3 = 0.0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) mine3 = (k == r) ? bx[k] : mine3;
                    dQdv[(int64_t)(16 + r) * n + i] = code == LOCAL_SUCCESS ? mine3 : 0.0;
                }
                if (dQdv && valid) dQdv[(int64_t)r * n + i] = 0.0;
            }
            if (dQdl && valid) {
                double mine = 0.0, mine2 = 0.0;
#pragma unroll
                for (int k = 0; k < 16; ++k) mine = (k == r) ? y[k] : mine;
#pragma unroll
                for (int k = 0; k < 4; ++k) mine2 = (k == r) ? y[16 + k] : mine2;
                dQdl[(int64_t)r * n + i] = mine;
                if (r < 4) dQdl[(int64_t)(16 + r) * n + i] = mine2;
            }
        }
        if (valid) {
            double mine = 0.0, mine2 = 0.0;
#pragma unroll
            for (int k = 0; k < 16; ++k) mine = (k == r) ? q[k] : mine;
#pragma unroll
            for (int k = 0; k < 4; ++k) mine2 = (k == r) ? q[16 + k] : mine2;
            Qs[(int64_t)r * n + i] = mine;
            if (r < 4) Qs[(int64_t)(16 + r) * n + i] = mine2;
            if (r == 0) {
                if (act) { // P_active = a ∂λ/∂F;  dP_active/du = a ∂²λ/∂F² + b ∂λ/∂F ⊗ ∂λ/∂F + c ∂λ/∂F ⊗ (∂²λ/∂F² : Ḟ)
                    const double fso = rdq20_overlap(p, lam), qq = q[17] + q[19];
                    act[act_stride * i] = tmax * qq * fso;
                    // rate-coupled: ∂P/∂Ḟ ∂Ḟ/∂u = q_v fso Tmax/Δt ∂λ⊗∂λ joins the symmetric coefficient (backward Euler: ∂Ḟ/∂u = 1/Δt)
                    act[act_stride * i + 1] = tmax * (qq * rdq20_overlap_slope(p, lam) + ql * fso) + (rate ? tmax * qv * fso / dt : 0.0);
                    if (rate) act[act_stride * i + 2] = tmax * qv * fso;
                }
                if (status) status[i] = code;
                if (code != LOCAL_SUCCESS && n_failed) atomicAdd(n_failed, 1ull);
            }
        }
    }
}

int launch_sarcomere_implicit(tb_device *dev, const double *params, double *d_Q, const double *d_Qknown, int64_t n, const double *d_stretch,
                              const double *d_velocity, const double *d_calcium, double stretch, double velocity, double calcium, double dt, double tol,
                              int max_iters, double tmax, double *d_dQdl, double *d_dQdv, double *d_act, int act_stride, int32_t *d_status, int64_t *n_failed,
                              int need_sens, const int32_t *d_cells, int nq, int64_t n_cells_listed)
{
    if (n_failed) TB_NO_CAPTURE(dev); // the failure count goes to the host
    const SarcomereInputs in{d_stretch, d_velocity, d_calcium, stretch, velocity, calcium};
    const int64_t n_work = d_cells ? n_cells_listed * nq : n;
    if (n_work == 0) { if (n_failed) *n_failed = 0; return TB_OK; }
    unsigned long long *cnt = (unsigned long long *)&dev->d_status->cell; // 8-byte scratch inside the status block
    TB_HIP(hipMemsetAsync(cnt, 0, sizeof *cnt, dev->stream));
    int64_t nb = (n_work + 15) / 16;
    const int64_t cap = (int64_t)dev->n_cu * 8;
    if (nb > cap) nb = cap;
    const RDQ20Params p = rdq20_params(params);
    if (need_sens) hipLaunchKernelGGL(k_sarcomere_implicit<true>, dim3((unsigned)nb), dim3(256), 0, dev->stream, p, in, d_Q, d_Qknown, n, dt, tol, max_iters, tmax,
                                      d_dQdl, d_dQdv, d_act, act_stride, d_status, cnt, d_cells, nq, n_work);
    else hipLaunchKernelGGL(k_sarcomere_implicit<false>, dim3((unsigned)nb), dim3(256), 0, dev->stream, p, in, d_Q, d_Qknown, n, dt, tol, max_iters, tmax,
                            d_dQdl, d_dQdv, d_act, act_stride, d_status, cnt, d_cells, nq, n_work);
    TB_HIP(hipGetLastError());
    if (n_failed) {
        unsigned long long h = 0;
        TB_HIP(hipMemcpyAsync(&h, cnt, sizeof h, hipMemcpyDeviceToHost, dev->stream));
        TB_SYNC_STREAM(dev);
        *n_failed = (int64_t)h;
    }
    return TB_OK;
}

int host_sarcomere_local_solve(const double *params, double *Q, const double *Qknown, double stretch, double velocity, double calcium, double dt, double tol,
                               int max_iters, double *dQdl, double *dQdv, int *iters, double *resnorm)
{
    const RDQ20Params p = rdq20_params(params);
    double q[20], qk[20];
    for (int k = 0; k < 20; ++k) { q[k] = Q[k]; qk[k] = Qknown[k]; }
    const int code = rdq20_local_solve_host(p, q, qk, stretch, velocity, calcium, dt, tol, max_iters, dQdl, iters, resnorm, dQdv);
    for (int k = 0; k < 20; ++k) Q[k] = q[k];
    return code;
}

// ∂rhs/∂u, ∂rhs/∂λ, ∂rhs/∂(dλ/dt): hand-derived (analytic != 0, what the kernel uses) or by forward mode (analytic == 0)
void host_sarcomere_derivatives(const double *params, const double *u, double stretch, double velocity, double calcium, int analytic, double *D, double *gl,
                                double *gv, double *rhs)
{
    const RDQ20Params p = rdq20_params(params);
    double q[20], Dm[20][20], l[20], v[20], f[20];
    for (int k = 0; k < 20; ++k) q[k] = u[k];
    if (analytic) rdq20_analytic_derivatives(p, q, stretch, velocity, calcium, Dm, l, v, f);
    else {
        for (int c = 0; c < 22; ++c) {
            D1 ud[20], dud[20];
            for (int k = 0; k < 20; ++k) ud[k] = {q[k], k == c ? 1.0 : 0.0};
            rdq20_rhs<D1>(p, ud, D1{stretch, c == 20 ? 1.0 : 0.0}, D1{velocity, c == 21 ? 1.0 : 0.0}, calcium, dud);
            for (int k = 0; k < 20; ++k) { if (c < 20) Dm[k][c] = dud[k].d; else if (c == 20) l[k] = dud[k].d; else v[k] = dud[k].d; f[k] = dud[k].v; }
        }
    }
    for (int k = 0; k < 20; ++k) { for (int c = 0; c < 20; ++c) D[20 * k + c] = Dm[k][c]; gl[k] = l[k]; gv[k] = v[k]; rhs[k] = f[k]; }
}

void host_sarcomere_eval(const double *params, const double *u, double stretch, double velocity, double calcium, double *du, double *tension,
                         double *stiffness)
{
    const RDQ20Params p = rdq20_params(params);
    double ul[RDQ20_NSTATES], dul[RDQ20_NSTATES];
    for (int k = 0; k < RDQ20_NSTATES; ++k) ul[k] = u[k];
    rdq20_rhs(p, ul, stretch, velocity, calcium, dul);
    if (du) for (int k = 0; k < RDQ20_NSTATES; ++k) du[k] = dul[k];
    if (tension) *tension = rdq20_tension(p, ul, stretch);
    if (stiffness) *stiffness = rdq20_stiffness(p, ul, stretch);
}

} // namespace tb
