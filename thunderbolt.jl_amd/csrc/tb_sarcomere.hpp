// Sarcomere models with internal state, evaluated pointwise: one header for the kernels and for the host evaluation the CPU tests pin.
//
// RDQ20-MF (Regazzoni–Dedè–Quarteroni 2020, mean-field variant): 16 occupancies of a regulatory-unit Markov chain over
// (T_left, T_centre, T_right, C_centre) ∈ {1,2}⁴ — flat index (TL−1) + 2(TC−1) + 4(TR−1) + 8(CC−1), Julia's column-major reshape —
// and 4 cross-bridge moments.  Restates sarcomere_rhs!(du, u, λ, dλdt, Ca, t, p::RDQ20MFModel)
// (src/modeling/solid/contraction.jl:385-453,532-583) exactly as written there, including the transposed use of the neighbour
// rates dT_L[TC, TL] / dT_R[TC, TR] (:432-436), which is what the reference's golden trajectory
// (test/data/trajectories/RDQ20-MF/transient-test.csv, test/test_sarcomere.jl:7-115) pins.
#pragma once
#include <cmath>
#ifndef TB_HD
#define TB_HD __host__ __device__ __forceinline__
#endif

namespace tbk {

constexpr int RDQ20_NSTATES = 20;
constexpr int RDQ20_NPARAMS = 17;
// field order of the reference struct (contraction.jl:337-369)
struct RDQ20Params {
    double LA, LM, LB, SL0, Q, Kd0, aKd, mu, gamma, Koff, Kbasic, r0, alpha, mu0_fP, mu1_fP, a_XB, eps_v;
};

// smooth_abs(x, ε) = x² / √(x² + ε²)  (src/utils.jl:580)
TB_HD double smooth_abs(double x, double e) { return x * x / ::sqrt(x * x + e * e); }

// γ^n for n ∈ {0, 1, 2}: the reference evaluates p.γ^n with an integer exponent (exact repeated multiplication)
TB_HD double rdq20_gpow(double g, int n) { return n == 0 ? 1.0 : (n == 1 ? g : g * g); }

// _rdq20_transition_rates (contraction.jl:385-397): a = TL−1, b = TC−1, c = TR−1, d = CC−1
TB_HD double rdq20_dT(const RDQ20Params &p, int a, int b, int c, int d)
{
    const int n = a + c; // permissive neighbours TL + TR − 2
    if (b == 1) return p.Kbasic * rdq20_gpow(p.gamma, 2 - n);
    if (d == 0) return p.Q * p.Kbasic * rdq20_gpow(p.gamma, n) / p.mu;
    return p.Q * p.Kbasic * rdq20_gpow(p.gamma, n);
}

// value + one directional derivative: the local Newton of the condensed mechanics differentiates the right-hand side by forward
// mode, one direction per lane (the reference: ForwardDiff.jacobian!, materials.jl:1448)
struct D1 { double v, d; };
TB_HD D1 operator+(D1 a, D1 b) { return {a.v + b.v, a.d + b.d}; }
TB_HD D1 operator-(D1 a, D1 b) { return {a.v - b.v, a.d - b.d}; }
TB_HD D1 operator-(D1 a) { return {-a.v, -a.d}; }
TB_HD D1 operator*(D1 a, D1 b) { return {a.v * b.v, a.d * b.v + a.v * b.d}; }
TB_HD D1 operator/(D1 a, D1 b) { const double q = a.v / b.v; return {q, (a.d - q * b.d) / b.v}; }
TB_HD D1 operator+(D1 a, double b) { return {a.v + b, a.d}; }
TB_HD D1 operator+(double a, D1 b) { return {a + b.v, b.d}; }
TB_HD D1 operator-(D1 a, double b) { return {a.v - b, a.d}; }
TB_HD D1 operator-(double a, D1 b) { return {a - b.v, -b.d}; }
TB_HD D1 operator*(D1 a, double b) { return {a.v * b, a.d * b}; }
TB_HD D1 operator*(double a, D1 b) { return {a * b.v, a * b.d}; }
TB_HD D1 operator/(D1 a, double b) { return {a.v / b, a.d / b}; }
TB_HD D1 operator/(double a, D1 b) { const double q = a / b.v; return {q, -q * b.d / b.v}; }
TB_HD D1 sqrt(D1 a) { const double r = ::sqrt(a.v); return {r, 0.5 * a.d / r}; }
TB_HD D1 smooth_abs(D1 x, double e) { return x * x / sqrt(x * x + e * e); }
TB_HD double val(double x) { return x; }
TB_HD double val(D1 x) { return x.v; }
template <class T> TB_HD T num(double c);
template <> TB_HD double num<double>(double c) { return c; }
template <> TB_HD D1 num<D1>(double c) { return {c, 0.0}; }

// permissivity and the permissive ↔ non-permissive rates the cross-bridge block reads from the regulatory units (contraction.jl:545-566)
template <class T> TB_HD void rdq20_xb_rates(const RDQ20Params &p, const T (&u)[RDQ20_NSTATES], T &perm, T &k_PN, T &k_NP)
{
#define RU(a, b, c, d) ((a) + 2 * (b) + 4 * (c) + 8 * (d))
    perm = num<T>(0.0);
    T flux_PN = num<T>(0.0), flux_NP = num<T>(0.0);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                perm = perm + u[RU(a, 1, c, d)];
                flux_PN = flux_PN + u[RU(a, 1, c, d)] * rdq20_dT(p, a, 1, c, d);
                flux_NP = flux_NP + u[RU(a, 0, c, d)] * rdq20_dT(p, a, 0, c, d);
            }
    k_PN = val(perm) >= 1e-12 ? flux_PN / perm : num<T>(0.0);
    k_NP = 1.0 - val(perm) >= 1e-12 ? flux_NP / (1.0 - perm) : num<T>(0.0);
#undef RU
}

template <class T>
TB_HD void rdq20_rhs(const RDQ20Params &p, const T (&u)[RDQ20_NSTATES], T lam, T dlam, double Ca, T (&du)[RDQ20_NSTATES])
{
#define RU(a, b, c, d) ((a) + 2 * (b) + 4 * (c) + 8 * (d))
    const T SL = p.SL0 * lam;
    const T dC1 = p.Koff / (p.Kd0 - p.aKd * (2.15 - SL)) * Ca;
    // dC[CC][TC] = [dC1 dC1; Koff Koff/μ]
    const T dC[2][2] = {{dC1, dC1}, {num<T>(p.Koff), num<T>(p.Koff / p.mu)}};
    T PhiT_C[16], PhiC_C[16];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int i = RU(a, b, c, d);
                    PhiT_C[i] = u[i] * rdq20_dT(p, a, b, c, d);
                    PhiC_C[i] = u[i] * dC[d][b];
                }
    // sums over CC, then over TR (left) / TL (right): the same association as sum(…; dims = 4) followed by dims = 3 / 1
    T su[8], sP[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { su[i] = u[i] + u[i + 8]; sP[i] = PhiT_C[i] + PhiT_C[i + 8]; }
    T rateL[2][2], rateR[2][2]; // rateL[TL][TC], rateR[TR][TC] as the reference builds them
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const T fl = sP[a + 2 * b] + sP[a + 2 * b + 4], pr = su[a + 2 * b] + su[a + 2 * b + 4];
            rateL[a][b] = val(pr) > 1e-12 ? fl / pr : num<T>(0.0);
        }
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const T fl = sP[2 * b + 4 * c] + sP[1 + 2 * b + 4 * c], pr = su[2 * b + 4 * c] + su[1 + 2 * b + 4 * c];
            rateR[c][b] = val(pr) > 1e-12 ? fl / pr : num<T>(0.0);
        }
    // ΦT_L = u · dT_L[TC, TL], ΦT_R = u · dT_R[TC, TR]  (indices as written in the reference)
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const T L_self = u[RU(a, b, c, d)] * rateL[b][a], L_flip = u[RU(1 - a, b, c, d)] * rateL[b][1 - a];
                    const T R_self = u[RU(a, b, c, d)] * rateR[b][c], R_flip = u[RU(a, b, 1 - c, d)] * rateR[b][1 - c];
                    du[RU(a, b, c, d)] = -L_self + L_flip - PhiT_C[RU(a, b, c, d)] + PhiT_C[RU(a, 1 - b, c, d)] - R_self + R_flip -
                                         PhiC_C[RU(a, b, c, d)] + PhiC_C[RU(a, b, c, 1 - d)];
                }
    // cross-bridges (contraction.jl:545-583)
    T perm, k_PN, k_NP;
    rdq20_xb_rates(p, u, perm, k_PN, k_NP);
    const T r = p.r0 + p.alpha * smooth_abs(dlam, p.eps_v);
    const T dP = r + k_PN, dN = r + k_NP;
    const T x1 = u[16], x2 = u[17], x3 = u[18], x4 = u[19];
    du[16] = -dP * x1 + k_NP * x3 + p.mu0_fP * perm;
    du[17] = dlam * x1 - dP * x2 + k_NP * x4 + p.mu1_fP * perm;
    du[18] = k_PN * x1 - dN * x3;
    du[19] = k_PN * x2 + dlam * x3 - dN * x4;
#undef RU
}

// fraction_single_overlap (contraction.jl:598-614)
TB_HD double rdq20_overlap(const RDQ20Params &p, double lam)
{
    const double SL = lam * p.SL0, LMh = (p.LM - p.LB) * 0.5;
    if (SL > p.LA && SL <= p.LM) return (SL - p.LA) / LMh;
    if (SL > p.LM && SL <= 2 * p.LA - p.LB) return (SL + p.LM - 2 * p.LA) * 0.5 / LMh;
    if (SL > 2 * p.LA - p.LB && SL <= 2 * p.LA + p.LB) return 1.0;
    if (SL > 2 * p.LA + p.LB && SL <= 2 * p.LA + p.LM) return (p.LM + 2 * p.LA - SL) * 0.5 / LMh;
    return 0.0;
}
// d fso / dλ of the branch the comparisons select (what forward-mode AD of fraction_single_overlap yields)
TB_HD double rdq20_overlap_slope(const RDQ20Params &p, double lam)
{
    const double SL = lam * p.SL0, LMh = (p.LM - p.LB) * 0.5;
    if (SL > p.LA && SL <= p.LM) return p.SL0 / LMh;
    if (SL > p.LM && SL <= 2 * p.LA - p.LB) return 0.5 * p.SL0 / LMh;
    if (SL > 2 * p.LA - p.LB && SL <= 2 * p.LA + p.LB) return 0.0;
    if (SL > 2 * p.LA + p.LB && SL <= 2 * p.LA + p.LM) return -0.5 * p.SL0 / LMh;
    return 0.0;
}
// compute_active_tension / compute_active_stiffness (contraction.jl:616-622): states 18, 20 / 17, 19 (1-based)
TB_HD double rdq20_tension(const RDQ20Params &p, const double (&u)[RDQ20_NSTATES], double lam) { return p.a_XB * (u[17] + u[19]) * rdq20_overlap(p, lam); }
TB_HD double rdq20_stiffness(const RDQ20Params &p, const double (&u)[RDQ20_NSTATES], double lam) { return p.a_XB * (u[16] + u[18]) * rdq20_overlap(p, lam); }

// ---- local problem of the condensed mechanics (materials.jl:1403-1497): backward Euler on the internal state at frozen stretch,
//      (Q − Q_known)/Δt = rhs(Q, λ, dλdt, Ca), by Newton with the forward-mode Jacobian; then the corrector dQ/dλ = J⁻¹ ∂rhs/∂λ
//      (:1556-1568, implicit function theorem).  Status codes follow the reference's LocalSolveReport retcodes.
enum { LOCAL_SUCCESS = 0, LOCAL_LINEAR_SOLVE_FAILED = 1, LOCAL_MAX_ITERS = 2, LOCAL_CONVERGENCE_FAILURE = 3, LOCAL_INFEASIBLE = 4 };

// dense host version (20×20, partial pivoting) — the device kernel distributes the same algebra over 16 lanes per point
inline bool dense_solve20(double (&A)[20][20], double (&b)[20])
{
    for (int k = 0; k < 20; ++k) {
        int piv = k;
        for (int i = k + 1; i < 20; ++i) if (std::fabs(A[i][k]) > std::fabs(A[piv][k])) piv = i;
        if (!(std::fabs(A[piv][k]) > 0.0)) return false;
        if (piv != k) { for (int j = 0; j < 20; ++j) { const double t = A[k][j]; A[k][j] = A[piv][j]; A[piv][j] = t; } const double t = b[k]; b[k] = b[piv]; b[piv] = t; }
        for (int i = k + 1; i < 20; ++i) {
            const double m = A[i][k] / A[k][k];
            for (int j = k; j < 20; ++j) A[i][j] -= m * A[k][j];
            b[i] -= m * b[k];
        }
    }
    for (int i = 19; i >= 0; --i) { double s2 = b[i]; for (int j = i + 1; j < 20; ++j) s2 -= A[i][j] * b[j]; b[i] = s2 / A[i][i]; }
    return true;
}
inline void rdq20_local_jacobian(const RDQ20Params &p, const double (&Q)[20], double lam, double dlam, double Ca, double dt, double (&J)[20][20], double (&rhs)[20])
{
    for (int c = 0; c < 20; ++c) {
        D1 u[20], du[20];
        for (int k = 0; k < 20; ++k) u[k] = {Q[k], k == c ? 1.0 : 0.0};
        rdq20_rhs<D1>(p, u, D1{lam, 0.0}, D1{dlam, 0.0}, Ca, du);
        for (int k = 0; k < 20; ++k) { J[k][c] = (k == c ? 1.0 / dt : 0.0) - du[k].d; rhs[k] = du[k].v; }
    }
}
// Q: initial guess in, solution out.  dQdl: dQ/dλ at the solution (may be nullptr).  Returns the status code.
inline int rdq20_local_solve_host(const RDQ20Params &p, double (&Q)[20], const double (&Qk)[20], double lam, double dlam, double Ca, double dt, double tol,
                                  int max_iters, double *dQdl, int *iters, double *resnorm, double *dQdv = nullptr)
{
    double J[20][20], f[20], R[20];
    double rn = 0.0;
    int it = 1;
    for (; it <= max_iters; ++it) {
        rdq20_local_jacobian(p, Q, lam, dlam, Ca, dt, J, f);
        rn = 0.0;
        for (int k = 0; k < 20; ++k) { R[k] = (Q[k] - Qk[k]) / dt - f[k]; rn += R[k] * R[k]; }
        rn = std::sqrt(rn);
        if (!dense_solve20(J, R)) { if (iters) *iters = it; if (resnorm) *resnorm = rn; return LOCAL_LINEAR_SOLVE_FAILED; }
        for (int k = 0; k < 20; ++k) Q[k] -= R[k];
        if (rn < tol) break;
        if (it == max_iters) { if (iters) *iters = it; if (resnorm) *resnorm = rn; return LOCAL_MAX_ITERS; }
        if (rn != rn) { if (iters) *iters = it; if (resnorm) *resnorm = rn; return LOCAL_CONVERGENCE_FAILURE; }
    }
    if (iters) *iters = it;
    if (resnorm) *resnorm = rn;
    for (int k = 0; k < 16; ++k) if (!(Q[k] >= 0.0)) return LOCAL_INFEASIBLE;
    if (dQdl) {
        rdq20_local_jacobian(p, Q, lam, dlam, Ca, dt, J, f);
        D1 u[20], du[20];
        for (int k = 0; k < 20; ++k) u[k] = {Q[k], 0.0};
        rdq20_rhs<D1>(p, u, D1{lam, 1.0}, D1{dlam, 0.0}, Ca, du);
        double g[20];
        for (int k = 0; k < 20; ++k) g[k] = du[k].d;
        if (!dense_solve20(J, g)) return LOCAL_LINEAR_SOLVE_FAILED;
        for (int k = 0; k < 20; ++k) dQdl[k] = g[k];
        if (dQdv) { // second corrector of the rate-coupled form: dQ/d(dλ/dt) (materials.jl:1715-1720)
            rdq20_local_jacobian(p, Q, lam, dlam, Ca, dt, J, f);
            rdq20_rhs<D1>(p, u, D1{lam, 0.0}, D1{dlam, 1.0}, Ca, du);
            for (int k = 0; k < 20; ++k) g[k] = du[k].d;
            if (!dense_solve20(J, g)) return LOCAL_LINEAR_SOLVE_FAILED;
            for (int k = 0; k < 20; ++k) dQdv[k] = g[k];
        }
    }
    return LOCAL_SUCCESS;
}

} // namespace tbk
