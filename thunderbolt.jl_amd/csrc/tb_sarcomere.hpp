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

// ---- the same right-hand side, differentiated by hand: what the 16-lanes-per-point kernel evaluates (one column per lane).
// The chain's nonlinearity sits in eight neighbour rates (flux / probability of the four (TL,TC) and the four (TR,TC) groups) and the
// cross-bridge block's in k_PN, k_NP; everything else is linear in the state.  Checked against the forward-mode columns above.
struct RDQ20Lin {
    double dT[16];                  // centre-unit transition rate of every state
    double dC[2][2];                // dC[CC][TC]
    double ddC1;                    // ∂dC1/∂λ
    double rateL[2][2], gLden[2][2]; // rateL[TL][TC] and 1/probability of the group (0 where the rate is switched off)
    double rateR[2][2], gRden[2][2]; // rateR[TR][TC]
    double perm, kPN, kNP, r, dr;   // cross-bridge block: permissivity, rates, r = r₀ + α|v|_ε and dr/dv
};
TB_HD void rdq20_linearise(const RDQ20Params &p, const double (&u)[RDQ20_NSTATES], double lam, double dlam, double Ca, RDQ20Lin &L)
{
    const double den = p.Kd0 - p.aKd * (2.15 - p.SL0 * lam);
    const double dC1 = p.Koff / den * Ca;
    L.dC[0][0] = dC1; L.dC[0][1] = dC1; L.dC[1][0] = p.Koff; L.dC[1][1] = p.Koff / p.mu;
    L.ddC1 = -p.Koff * Ca * p.aKd * p.SL0 / (den * den);
    double su[8], sP[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) L.dT[i] = rdq20_dT(p, i & 1, (i >> 1) & 1, (i >> 2) & 1, (i >> 3) & 1);
#pragma unroll
    for (int i = 0; i < 8; ++i) { su[i] = u[i] + u[i + 8]; sP[i] = u[i] * L.dT[i] + u[i + 8] * L.dT[i + 8]; }
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const double fl = sP[a + 2 * b] + sP[a + 2 * b + 4], pr = su[a + 2 * b] + su[a + 2 * b + 4];
            const bool on = pr > 1e-12;
            L.rateL[a][b] = on ? fl / pr : 0.0;
            L.gLden[a][b] = on ? 1.0 / pr : 0.0;
        }
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const double fl = sP[2 * b + 4 * c] + sP[1 + 2 * b + 4 * c], pr = su[2 * b + 4 * c] + su[1 + 2 * b + 4 * c];
            const bool on = pr > 1e-12;
            L.rateR[c][b] = on ? fl / pr : 0.0;
            L.gRden[c][b] = on ? 1.0 / pr : 0.0;
        }
    rdq20_xb_rates<double>(p, u, L.perm, L.kPN, L.kNP);
    const double e2 = p.eps_v * p.eps_v, x2 = dlam * dlam, sq = ::sqrt(x2 + e2);
    L.r = p.r0 + p.alpha * (x2 / sq);
    L.dr = p.alpha * dlam * (x2 + 2.0 * e2) / ((x2 + e2) * sq);   // d/dx [x²/√(x²+ε²)]
}
// m[i][j] for run-time i, j without indexing a register array by a run-time value
TB_HD double pick22(const double (&m)[2][2], int i, int j) { return i ? (j ? m[1][1] : m[1][0]) : (j ? m[0][1] : m[0][0]); }
// rhs_i for one row i (0…19), given the linearisation data
TB_HD double rdq20_rhs_row(const RDQ20Params &p, const RDQ20Lin &L, const double (&u)[RDQ20_NSTATES], double dlam, int i)
{
    if (i < 16) {
        const int a = i & 1, b = (i >> 1) & 1, c = (i >> 2) & 1, d = (i >> 3) & 1;
        return -u[i] * L.rateL[b][a] + u[i ^ 1] * L.rateL[b][1 - a] - u[i] * L.dT[i] + u[i ^ 2] * L.dT[i ^ 2] - u[i] * L.rateR[b][c] + u[i ^ 4] * L.rateR[b][1 - c] -
               u[i] * L.dC[d][b] + u[i ^ 8] * L.dC[1 - d][b];
    }
    const double dP = L.r + L.kPN, dN = L.r + L.kNP, x1 = u[16], x2 = u[17], x3 = u[18], x4 = u[19];
    if (i == 16) return -dP * x1 + L.kNP * x3 + p.mu0_fP * L.perm;
    if (i == 17) return dlam * x1 - dP * x2 + L.kNP * x4 + p.mu1_fP * L.perm;
    if (i == 18) return L.kPN * x1 - dN * x3;
    return L.kPN * x2 + dlam * x3 - dN * x4;
}
// all twenty rows at once (compile-time indices only)
TB_HD void rdq20_rhs_rows(const RDQ20Params &p, const RDQ20Lin &L, const double (&u)[RDQ20_NSTATES], double dlam, double (&f)[RDQ20_NSTATES])
{
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int a = i & 1, b = (i >> 1) & 1, c = (i >> 2) & 1, d = (i >> 3) & 1;
        f[i] = -u[i] * L.rateL[b][a] + u[i ^ 1] * L.rateL[b][1 - a] - u[i] * L.dT[i] + u[i ^ 2] * L.dT[i ^ 2] - u[i] * L.rateR[b][c] + u[i ^ 4] * L.rateR[b][1 - c] -
               u[i] * L.dC[d][b] + u[i ^ 8] * L.dC[1 - d][b];
    }
    const double dP = L.r + L.kPN, dN = L.r + L.kNP, x1 = u[16], x2 = u[17], x3 = u[18], x4 = u[19];
    f[16] = -dP * x1 + L.kNP * x3 + p.mu0_fP * L.perm;
    f[17] = dlam * x1 - dP * x2 + L.kNP * x4 + p.mu1_fP * L.perm;
    f[18] = L.kPN * x1 - dN * x3;
    f[19] = L.kPN * x2 + dlam * x3 - dN * x4;
}
// column c (0…15) of ∂rhs/∂u: col[0…15] chain rows, col[16…19] cross-bridge rows
TB_HD void rdq20_rhs_column(const RDQ20Params &p, const RDQ20Lin &L, const double (&u)[RDQ20_NSTATES], int c, double (&col)[RDQ20_NSTATES])
{
    const int a = c & 1, b = (c >> 1) & 1, cc = (c >> 2) & 1, d = (c >> 3) & 1;
#pragma unroll
    for (int i = 0; i < RDQ20_NSTATES; ++i) col[i] = 0.0;
    const double dTc = rdq20_dT(p, a, b, cc, d);
    const double rL_ba = pick22(L.rateL, b, a), rR_bc = pick22(L.rateR, b, cc), dC_db = pick22(L.dC, d, b);
    // explicit occurrences of u_c
    const double diag = -(rL_ba + dTc + rR_bc + dC_db);
    const double gL = (dTc - pick22(L.rateL, a, b)) * pick22(L.gLden, a, b);   // ∂ rateL[TL=a][TC=b] / ∂u_c
    const double gR = (dTc - pick22(L.rateR, cc, b)) * pick22(L.gRden, cc, b); // ∂ rateR[TR=cc][TC=b] / ∂u_c
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int ai = i & 1, bi = (i >> 1) & 1, ci = (i >> 2) & 1;
        double v = 0.0;
        v += (i == c) ? diag : 0.0;
        v += (i == (c ^ 1)) ? rL_ba : 0.0;
        v += (i == (c ^ 2)) ? dTc : 0.0;
        v += (i == (c ^ 4)) ? rR_bc : 0.0;
        v += (i == (c ^ 8)) ? dC_db : 0.0;
        // through the neighbour rates: row i reads rateL[b_i][a_i] (self) and rateL[b_i][1 − a_i] (flipped left neighbour) …
        if (bi == a) v += (ai == b) ? -u[i] * gL : u[i ^ 1] * gL;
        // … and rateR[b_i][c_i] (self), rateR[b_i][1 − c_i] (flipped right neighbour)
        if (bi == cc) v += (ci == b) ? -u[i] * gR : u[i ^ 4] * gR;
        col[i] = v;
    }
    // cross-bridge rows through the permissivity and the two exchange rates
    const double dperm = b ? 1.0 : 0.0;
    const double dkPN = L.perm >= 1e-12 ? ((b ? dTc : 0.0) - L.kPN * dperm) / L.perm : 0.0;
    const double dkNP = 1.0 - L.perm >= 1e-12 ? ((b ? 0.0 : dTc) + L.kNP * dperm) / (1.0 - L.perm) : 0.0;
    const double x1 = u[16], x2 = u[17], x3 = u[18], x4 = u[19];
    col[16] = -dkPN * x1 + dkNP * x3 + p.mu0_fP * dperm;
    col[17] = -dkPN * x2 + dkNP * x4 + p.mu1_fP * dperm;
    col[18] = dkPN * x1 - dkNP * x3;
    col[19] = dkPN * x2 - dkNP * x4;
}
// ∂rhs/∂λ (only through the calcium binding rate) and ∂rhs/∂(dλ/dt) (only the cross-bridge rows)
TB_HD double rdq20_rhs_dlam_row(const RDQ20Lin &L, const double (&u)[RDQ20_NSTATES], int i)
{
    if (i >= 16) return 0.0;
    const int d = (i >> 3) & 1;
    return d == 0 ? -u[i] * L.ddC1 : u[i ^ 8] * L.ddC1;
}
TB_HD void rdq20_rhs_dvel(const RDQ20Lin &L, const double (&u)[RDQ20_NSTATES], double (&g)[4])
{
    g[0] = -L.dr * u[16]; g[1] = u[16] - L.dr * u[17]; g[2] = -L.dr * u[18]; g[3] = u[18] - L.dr * u[19];
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
// hand-derived counterpart of rdq20_local_jacobian's derivative part: ∂rhs/∂u (20×20), ∂rhs/∂λ, ∂rhs/∂(dλ/dt), and rhs itself
inline void rdq20_analytic_derivatives(const RDQ20Params &p, const double (&Q)[20], double lam, double dlam, double Ca, double (&D)[20][20], double (&gl)[20],
                                       double (&gv)[20], double (&rhs)[20])
{
    RDQ20Lin L;
    rdq20_linearise(p, Q, lam, dlam, Ca, L);
    for (int c = 0; c < 16; ++c) {
        double col[20];
        rdq20_rhs_column(p, L, Q, c, col);
        for (int k = 0; k < 20; ++k) D[k][c] = col[k];
    }
    const double dP = L.r + L.kPN, dN = L.r + L.kNP;
    const double XB[4][4] = {{-dP, 0, L.kNP, 0}, {dlam, -dP, 0, L.kNP}, {L.kPN, 0, -dN, 0}, {0, L.kPN, dlam, -dN}};
    for (int k = 0; k < 20; ++k) for (int c = 16; c < 20; ++c) D[k][c] = k >= 16 ? XB[k - 16][c - 16] : 0.0;
    double g4[4];
    rdq20_rhs_dvel(L, Q, g4);
    for (int k = 0; k < 20; ++k) { gl[k] = rdq20_rhs_dlam_row(L, Q, k); gv[k] = k >= 16 ? g4[k - 16] : 0.0; rhs[k] = rdq20_rhs_row(p, L, Q, dlam, k); }
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
