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

TB_HD void rdq20_rhs(const RDQ20Params &p, const double (&u)[RDQ20_NSTATES], double lam, double dlam, double Ca, double (&du)[RDQ20_NSTATES])
{
#define RU(a, b, c, d) ((a) + 2 * (b) + 4 * (c) + 8 * (d))
    const double SL = p.SL0 * lam;
    const double dC1 = p.Koff / (p.Kd0 - p.aKd * (2.15 - SL)) * Ca;
    // dC[CC][TC] = [dC1 dC1; Koff Koff/μ]
    const double dC[2][2] = {{dC1, dC1}, {p.Koff, p.Koff / p.mu}};
    double PhiT_C[16], PhiC_C[16];
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
    double su[8], sP[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { su[i] = u[i] + u[i + 8]; sP[i] = PhiT_C[i] + PhiT_C[i + 8]; }
    double rateL[2][2], rateR[2][2]; // rateL[TL][TC], rateR[TR][TC] as the reference builds them
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const double fl = sP[a + 2 * b] + sP[a + 2 * b + 4], pr = su[a + 2 * b] + su[a + 2 * b + 4];
            rateL[a][b] = pr > 1e-12 ? fl / pr : 0.0;
        }
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const double fl = sP[2 * b + 4 * c] + sP[1 + 2 * b + 4 * c], pr = su[2 * b + 4 * c] + su[1 + 2 * b + 4 * c];
            rateR[c][b] = pr > 1e-12 ? fl / pr : 0.0;
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
                    const double L_self = u[RU(a, b, c, d)] * rateL[b][a], L_flip = u[RU(1 - a, b, c, d)] * rateL[b][1 - a];
                    const double R_self = u[RU(a, b, c, d)] * rateR[b][c], R_flip = u[RU(a, b, 1 - c, d)] * rateR[b][1 - c];
                    du[RU(a, b, c, d)] = -L_self + L_flip - PhiT_C[RU(a, b, c, d)] + PhiT_C[RU(a, 1 - b, c, d)] - R_self + R_flip -
                                         PhiC_C[RU(a, b, c, d)] + PhiC_C[RU(a, b, c, 1 - d)];
                }
    // cross-bridges (contraction.jl:545-583)
    double perm = 0.0, flux_PN = 0.0, flux_NP = 0.0;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                perm += u[RU(a, 1, c, d)];
                flux_PN += PhiT_C[RU(a, 1, c, d)];
                flux_NP += PhiT_C[RU(a, 0, c, d)];
            }
    const double k_PN = perm >= 1e-12 ? flux_PN / perm : 0.0;
    const double k_NP = 1.0 - perm >= 1e-12 ? flux_NP / (1.0 - perm) : 0.0;
    const double r = p.r0 + p.alpha * smooth_abs(dlam, p.eps_v);
    const double dP = r + k_PN, dN = r + k_NP;
    const double x1 = u[16], x2 = u[17], x3 = u[18], x4 = u[19];
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
// compute_active_tension / compute_active_stiffness (contraction.jl:616-622): states 18, 20 / 17, 19 (1-based)
TB_HD double rdq20_tension(const RDQ20Params &p, const double (&u)[RDQ20_NSTATES], double lam) { return p.a_XB * (u[17] + u[19]) * rdq20_overlap(p, lam); }
TB_HD double rdq20_stiffness(const RDQ20Params &p, const double (&u)[RDQ20_NSTATES], double lam) { return p.a_XB * (u[16] + u[18]) * rdq20_overlap(p, lam); }

} // namespace tbk
