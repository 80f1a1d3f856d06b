// tb_energy.hpp — the reference's strain-energy functions Ψ(F) (src/modeling/solid/energies.jl) written once over a generic
// number type, and a scalar hyper-dual number for differentiating them on the device.
//
// The reference obtains P = ∂Ψ/∂F and 𝔸 = ∂²Ψ/∂F² of *every* material by forward-mode AD (Tensors.gradient / Tensors.hessian,
// src/modeling/solid/materials.jl:1025-1040).  The Holzapfel–Ogden model has hand-derived derivatives here (tb_material.hpp, the fast
// path); all other energies go through the same idea the reference uses, mapped to the GPU: one lane per pair (m, n) of components of
// F evaluates Ψ on hyper-dual numbers  x = v + a ε₁ + b ε₂ + ab ε₁ε₂  (ε₁² = ε₂² = 0) seeded with ε₁ on F_m and ε₂ on F_n;
// then Ψ.a = P_m, Ψ.ab = 𝔸_mn = 𝔸_nm.  45 lanes per quadrature point cover the symmetric 9×9 tangent, each carrying four doubles
// per intermediate — no 9×9 tangent in anybody's registers.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

namespace tbk {

#ifndef TB_HD
#define TB_HD __host__ __device__ __forceinline__
#endif

struct HD {
    double v, a, b, ab;
};
TB_HD HD hd_c(double c) { return HD{c, 0.0, 0.0, 0.0}; }
TB_HD HD operator+(HD x, HD y) { return HD{x.v + y.v, x.a + y.a, x.b + y.b, x.ab + y.ab}; }
TB_HD HD operator-(HD x, HD y) { return HD{x.v - y.v, x.a - y.a, x.b - y.b, x.ab - y.ab}; }
TB_HD HD operator-(HD x) { return HD{-x.v, -x.a, -x.b, -x.ab}; }
TB_HD HD operator+(HD x, double c) { x.v += c; return x; }
TB_HD HD operator-(HD x, double c) { x.v -= c; return x; }
TB_HD HD operator+(double c, HD x) { x.v += c; return x; }
TB_HD HD operator-(double c, HD x) { return HD{c - x.v, -x.a, -x.b, -x.ab}; }
TB_HD HD operator*(HD x, double c) { return HD{x.v * c, x.a * c, x.b * c, x.ab * c}; }
TB_HD HD operator*(double c, HD x) { return x * c; }
TB_HD HD operator/(HD x, double c) { return x * (1.0 / c); }
TB_HD HD operator*(HD x, HD y)
{
    return HD{x.v * y.v, x.a * y.v + x.v * y.a, x.b * y.v + x.v * y.b, x.ab * y.v + x.a * y.b + x.b * y.a + x.v * y.ab};
}
// f(u) with f, f', f'' given at u.v
TB_HD HD hd_chain(HD u, double f, double d1, double d2) { return HD{f, d1 * u.a, d1 * u.b, d1 * u.ab + d2 * u.a * u.b}; }
TB_HD HD exp(HD u) { const double e = ::exp(u.v); return hd_chain(u, e, e, e); }
TB_HD HD log(HD u) { const double i = 1.0 / u.v; return hd_chain(u, ::log(u.v), i, -i * i); }
TB_HD HD sqrt(HD u) { const double s = ::sqrt(u.v); return hd_chain(u, s, 0.5 / s, -0.25 / (s * u.v)); }
TB_HD HD cbrt(HD u) { const double c = ::cbrt(u.v); return hd_chain(u, c, c / (3.0 * u.v), -2.0 * c / (9.0 * u.v * u.v)); }
TB_HD HD inv(HD u) { const double i = 1.0 / u.v; return hd_chain(u, i, -i * i, 2.0 * i * i * i); }
TB_HD HD operator/(HD x, HD y) { return x * inv(y); }
TB_HD HD operator/(double c, HD y) { return inv(y) * c; }
// u^p for real p (u.v ≥ 0; the derivatives at u.v = 0 are taken as 0 when their exponent is positive, like 0.0^q in Julia)
TB_HD HD pow(HD u, double p)
{
    const double f = ::pow(u.v, p);
    const double d1 = p == 0.0 ? 0.0 : p * ::pow(u.v, p - 1.0);
    const double d2 = (p == 0.0 || p == 1.0) ? 0.0 : p * (p - 1.0) * ::pow(u.v, p - 2.0);
    return hd_chain(u, f, d1, d2);
}
// the same vocabulary for plain doubles, so that an energy written once serves both
TB_HD double hd_value(double x) { return x; }
TB_HD double hd_value(HD x) { return x.v; }
TB_HD double inv(double x) { return 1.0 / x; }
TB_HD double exp(double x) { return ::exp(x); }   // the HD overloads above would otherwise hide the global ones inside tbk
TB_HD double log(double x) { return ::log(x); }
TB_HD double sqrt(double x) { return ::sqrt(x); }
TB_HD double cbrt(double x) { return ::cbrt(x); }
TB_HD double pow(double x, double p) { return ::pow(x, p); }

// ---- material description that travels by value into the kernels (tb_material of the ABI, lowered) ----
enum { EN_HOLZAPFEL_OGDEN = 0, EN_NULL = 1, EN_BIO_NEOHOOKEAN = 2, EN_TI_NEOHOOKEAN = 3, EN_LIN_YIN_PASSIVE = 4, EN_LIN_YIN_ACTIVE = 5,
       EN_HUMPHREY_STRUMPF_YIN = 6, EN_LINEAR_SPRING = 7, EN_GUCCIONE_1991 = 8 };
enum { PEN_SIMPLE = 0, PEN_NULL = 1, PEN_HARTMANN_NEFF_1 = 2, PEN_HARTMANN_NEFF_2 = 3, PEN_HARTMANN_NEFF_3 = 4 };

enum { HILL_NONE = 0, HILL_GENERALIZED = 1, HILL_EXTENDED = 2 };
enum { ACT_SIMPLE_ACTIVE_SPRING = 100 };                        // active spring that is not an ActiveMaterialAdapter over an energy
enum { ADG_GMK = 0, ADG_GMK_INCOMPRESSIBLE = 1, ADG_RLRSQ = 2 }; // active deformation gradient models, active.jl:23-96
enum { SARC_PELCE_SUN_LANGEVELD = 0, SARC_CONSTANT_STRETCH = 1 };

struct EnergyParams {
    int energy, penalty;
    double p[9];  // energy parameters in the field order of the reference struct
    double u[3];  // penalty parameters: β, a, b
    double Ta;    // per point: active tension Tmax·𝓝 (ActiveStressModel + SimpleActiveStress, + Ta·‖F f₀‖) — or, with a Hill framework,
                  // the calcium-driven state of the sarcomere model
    // GeneralizedHillModel / ExtendedHillModel (materials.jl:1042-1190): W = W_passive(F) + [𝓝] W_active(F·Fᵃ⁻¹)
    int hill, act_energy, act_penalty, adg, sarc;
    double ap[9], au[3]; // active spring: energy + penalty parameters (ACT_SIMPLE_ACTIVE_SPRING: ap[0] = aᶠ)
    double kappa;        // RLRSQ sheetlet part
    double sp[2];        // PelceSunLangeveld1995: β, λᵃₘₐₓ; ConstantStretch: λ
    // PrestressedMechanicalModel (materials.jl:781-900): W(F) = W_inner(F·F₀⁻¹); G = F₀⁻¹ row-major, used when prestressed != 0
    int prestressed;
    double G[9];
};

// U(I₃), energies.jl:13-87 (I₃ < 0 → NaN there; a cell with detJ ≤ 0 is reported separately)
template <class T> TB_HD T penalty_U(const EnergyParams &m, T I3)
{
    const double beta = m.u[0], a = m.u[1], b = m.u[2];
    switch (m.penalty) {
    case PEN_NULL: return I3 * 0.0;
    case PEN_HARTMANN_NEFF_1: return beta * pow(pow(I3, b) + inv(pow(I3, b)) - 2.0, a);  // β (I₃ᵇ + I₃⁻ᵇ − 2)ᵃ
    case PEN_HARTMANN_NEFF_2: return beta * pow(sqrt(I3) - 1.0, a);                         // β (√I₃ − 1)ᵃ
    case PEN_HARTMANN_NEFF_3: { const T lj = log(sqrt(I3)); return beta * (I3 - 2.0 * lj + 4.0 * (lj * lj) - 1.0); }
    default: return beta * (I3 - 1.0 - 2.0 * log(sqrt(I3)));                                // SimpleCompressionPenalty
    }
}

// Ψ(F; f₀, s₀, n₀).  Every branch follows the expression of the reference function it cites.
// Ta: the per-point active tension / calcium state (EnergyParams::Ta of a uniform state).  It travels as an argument of its own so that the parameter block
// stays uniform over the lanes (a lane-private copy of the 330-byte block lived in scratch memory)
// EN ≥ 0: the energy is known at compile time (the other branches vanish: the point kernel of the split linearisation is instantiated per energy — with the
// run-time switch its loop body needs more than the 512 registers of a lone wave)
template <class T, int EN = -1, bool HILL = false /* EN ≥ 0 inside a Hill framework: Ta carries the calcium, there is no active-stress term */>
TB_HD T energy_psi(const EnergyParams &m, const T (&F)[3][3], const double (&f0)[3], const double (&s0)[3], const double (&n0)[3], double Ta)
{
    T C[3][3]; // C = tdot(F) = FᵀF
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = i; j < 3; ++j) { C[i][j] = F[0][i] * F[0][j] + F[1][i] * F[1][j] + F[2][i] * F[2][j]; C[j][i] = C[i][j]; }
    const T I1 = C[0][0] + C[1][1] + C[2][2];
    const T I3 = C[0][0] * (C[1][1] * C[2][2] - C[1][2] * C[2][1]) - C[0][1] * (C[1][0] * C[2][2] - C[1][2] * C[2][0]) +
                 C[0][2] * (C[1][0] * C[2][1] - C[1][1] * C[2][0]);
    auto quad = [&](const double (&x)[3], const double (&y)[3]) { // x·C·y
        T r = C[0][0] * (x[0] * y[0]);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (i || j) r = r + C[i][j] * (x[i] * y[j]);
        return r;
    };
    const double *p = m.p;
    T psi = I1 * 0.0;
    switch (EN >= 0 ? EN : m.energy) {
    case EN_NULL: break;                                                                          // energies.jl:6-7
    case EN_BIO_NEOHOOKEAN: psi = p[0] * (I1 * inv(cbrt(I3)) - 3.0) + penalty_U(m, I3); break;    // :461-473
    case EN_TI_NEOHOOKEAN: {                                                                      // :93-128
        T I1b = I1 * inv(cbrt(I3));
        const double dv = hd_value(I1b) - 3.0;
        if (-1e-8 < dv && dv < 0.0) I1b = I1b * 0.0 + 3.0; // the reference's hotfix: a plain 3.0, derivatives gone
        psi = p[2] * pow(I1b - 3.0, p[0]) + penalty_U(m, I3);
        const T I4 = quad(f0, f0);
        if (hd_value(I4) >= 1.0) psi = psi + p[3] * pow(I4 - 1.0, p[1]);
        break;
    }
    case EN_LIN_YIN_PASSIVE: {                                                                    // :178-198
        const T i1 = I1 - 3.0, i4 = quad(f0, f0) - 1.0;
        const T Q = p[1] * (i1 * i1) + p[2] * (i1 * i4) + p[3] * (i4 * i4);
        psi = p[0] * (exp(Q) - 1.0) + penalty_U(m, I3);
        break;
    }
    case EN_LIN_YIN_ACTIVE: {                                                                     // :207-226
        const T i1 = I1 - 3.0, i4 = quad(f0, f0) - 1.0;
        psi = p[0] + p[1] * (i1 * i4) + p[2] * (i1 * i1) + p[3] * (i4 * i4) + p[4] * i1 + p[5] * i4 + penalty_U(m, I3);
        break;
    }
    case EN_HUMPHREY_STRUMPF_YIN: {                                                               // :235-252
        const T l = sqrt(quad(f0, f0)) - 1.0, i1 = I1 - 3.0;
        psi = p[0] * (l * l) + p[1] * (l * l * l) + p[2] * (l * i1) + p[3] * (i1 * i1) + penalty_U(m, I3);
        break;
    }
    case EN_LINEAR_SPRING: { const T i4 = quad(f0, f0) - 1.0; psi = (0.5 * p[0]) * (i4 * i4); break; } // :261-275 (mpU not applied there)
    case EN_GUCCIONE_1991: {                                                                      // :284-330, E = (C − I)/2
        auto Eq = [&](const double (&x)[3], const double (&y)[3]) { return 0.5 * (quad(x, y) - (x[0] * y[0] + x[1] * y[1] + x[2] * y[2])); };
        const T Eff = Eq(f0, f0), Ess = Eq(s0, s0), Enn = Eq(n0, n0), Efs = Eq(f0, s0), Esf = Eq(s0, f0), Esn = Eq(s0, n0), Ens = Eq(n0, s0),
                Efn = Eq(f0, n0), Enf = Eq(n0, f0);
        const T Q = p[1] * (Eff * Eff) + p[2] * (Ess * Ess) + p[3] * (Enn * Enn) + p[4] * (Ens * Ens + Esn * Esn) + p[5] * (Efs * Efs + Esf * Esf) +
                    p[6] * (Efn * Efn + Enf * Enf);
        psi = (0.5 * p[0]) * exp(Q) + penalty_U(m, I3);
        break;
    }
    default: {                                                                                    // Holzapfel–Ogden 2009, :136-168
        const T I1b = I1 * inv(cbrt(I3));
        const T I4f = quad(f0, f0), I4s = quad(s0, s0);
        const T I8 = 0.5 * (quad(f0, s0) + quad(s0, f0));
        psi = (p[0] / (2.0 * p[1])) * (exp(p[1] * (I1b - 3.0)) - 1.0) + (p[6] / (2.0 * p[7])) * (exp(p[7] * (I8 * I8)) - 1.0) + penalty_U(m, I3);
        if (hd_value(I4f) >= 1.0) { const T d = I4f - 1.0; psi = psi + (p[2] / (2.0 * p[3])) * (exp(p[3] * (d * d)) - 1.0); }
        if (hd_value(I4s) >= 1.0) { const T d = I4s - 1.0; psi = psi + (p[4] / (2.0 * p[5])) * (exp(p[5] * (d * d)) - 1.0); }
    }
    }
    if (!HILL && Ta != 0.0 && (EN >= 0 || m.hill == HILL_NONE)) psi = psi + Ta * sqrt(quad(f0, f0)); // active stress: ∂(Ta‖F f₀‖)/∂F (materials.jl:1200-1266, active.jl:100-113)
    return psi;
}

// compute_λᵃ of the steady-state sarcomere models (contraction.jl:302-320)
TB_HD double sarcomere_lambda_a(const EnergyParams &m, double Ca)
{
    if (m.sarc == SARC_CONSTANT_STRETCH) return m.sp[0];
    const double f = Ca > 0.0 ? 0.5 + atan(m.sp[0] * ::log(Ca)) / 3.141592653589793 : 0.0;
    return 1.0 / (1.0 + f * (1.0 / m.sp[1] - 1.0));
}

// Total energy of the material at F: passive part, plus — in the Hill frameworks — the active spring evaluated at Fᵉ = F·Fᵃ⁻¹.
// All three active deformation gradients of the reference are diagonal in the (orthonormal) frame, Fᵃ = a f⊗f + b s⊗s + c n⊗n, so
// Fᵃ⁻¹ is known in closed form and ActiveMaterialAdapter's rotated frame Fᵃf₀/‖Fᵃf₀‖ … is the frame itself (active.jl:8-21).
// HA ≥ 0 (with EN ≥ 0): a Hill framework whose active spring is the energy HA (an ActiveMaterialAdapter over it) or, HA = ACT_SIMPLE_ACTIVE_SPRING, the
// SimpleActiveSpring — passive energy and active spring both fixed at compile time, so that the instance keeps its registers (the run-time form of
// k_mech_points holds 1.3 KB in scratch).  HA = −1: no Hill framework at compile time.
template <class T, int EN = -1, int HA = -1> TB_HD T material_psi(const EnergyParams &m, const T (&F)[3][3], const double (&f0)[3], const double (&s0)[3], const double (&n0)[3], double Ta)
{
    constexpr bool HILLCT = EN >= 0 && HA >= 0;
    T psi = energy_psi<T, EN, HILLCT>(m, F, f0, s0, n0, Ta);
    if constexpr (EN >= 0 && !HILLCT) return psi; // compile-time energies of plain materials (no Hill framework, no prestress)
    if (!HILLCT && m.hill == HILL_NONE) return psi;
    const double Ca = Ta;
    const double la = sarcomere_lambda_a(m, Ca);
    double a = la, b = 1.0, c = 1.0;                                              // GMK: I + (λᵃ − 1) f⊗f
    if (m.adg == ADG_GMK_INCOMPRESSIBLE) { b = c = 1.0 / ::sqrt(la); }            // λᵃ f⊗f + λᵃ^{-1/2} (s⊗s + n⊗n)
    else if (m.adg == ADG_RLRSQ) { b = 1.0 + m.kappa * (la - 1.0); c = 1.0 / (b * la); }
    double Ai[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) Ai[i][j] = f0[i] * f0[j] / a + s0[i] * s0[j] / b + n0[i] * n0[j] / c;
    T Fe[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) Fe[i][j] = F[i][0] * Ai[0][j] + F[i][1] * Ai[1][j] + F[i][2] * Ai[2][j];
    T pa;
    if ((HILLCT && HA == ACT_SIMPLE_ACTIVE_SPRING) || (!HILLCT && m.act_energy == ACT_SIMPLE_ACTIVE_SPRING)) { // aᶠ/2 (f₀·Cᵉ f₀ − 1)², energies.jl:334-347
        T i4 = F[0][0] * 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) { const T v = Fe[k][0] * f0[0] + Fe[k][1] * f0[1] + Fe[k][2] * f0[2]; i4 = i4 + v * v; }
        const T d = i4 - 1.0;
        pa = (0.5 * m.ap[0]) * (d * d);
    } else {
        EnergyParams act{};
        act.energy = m.act_energy; act.penalty = m.act_penalty;
#pragma unroll
        for (int k = 0; k < 9; ++k) act.p[k] = m.ap[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) act.u[k] = m.au[k];
        pa = energy_psi<T, (HILLCT && HA != ACT_SIMPLE_ACTIVE_SPRING) ? HA : -1>(act, Fe, f0, s0, n0, 0.0);
    }
    return psi + (m.hill == HILL_EXTENDED ? Ca : 1.0) * pa;                        // 𝓝(state, …) = state for steady-state sarcomeres
}

// pair index pr ∈ [0, 45) ↔ components (mm ≤ nn) of F (row-major 0…8)
TB_HD void pair_components(int pr, int &mm, int &nn)
{
    int r = 0, left = pr;
    while (left >= 9 - r) { left -= 9 - r; ++r; }
    mm = r; nn = r + left;
}

// Ψ, P_mm and 𝔸_(mm,nn) at F by one hyper-dual evaluation
TB_HD HD energy_pair(const EnergyParams &m, const double *F9, int mm, int nn, const double (&f0)[3], const double (&s0)[3], const double (&n0)[3], double Ta)
{
    HD F[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int e = 3 * i + j;
            F[i][j] = HD{F9[e], e == mm ? 1.0 : 0.0, e == nn ? 1.0 : 0.0, 0.0};
        }
    if (m.prestressed) { // Fᵉ = F·F₀⁻¹: differentiating through the product gives P = Pᵉ·F₀⁻ᵀ and 𝔸_ijkl = 𝔸ᵉ_imkn G_jm G_ln
        HD Fe[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) Fe[i][j] = F[i][0] * m.G[j] + F[i][1] * m.G[3 + j] + F[i][2] * m.G[6 + j];
        return material_psi<HD>(m, Fe, f0, s0, n0, Ta);
    }
    return material_psi<HD>(m, F, f0, s0, n0, Ta);
}

// The same evaluation with DIRECTIONAL seeds: ε₁ along da, ε₂ along db (nine components each, row-major like F).  Ψ.a = P : da, Ψ.ab = da : 𝔸 : db.
// With da = e_c ⊗ J⁻¹[s][·] and db = e_d ⊗ J⁻¹[u][·] (F = I + Ĥ·J⁻¹ differentiated in Ĥ) these are the pulled-back stress P̂[c][s] and tangent
// Â[c][s][d][u] of the sum-factorised kernels, without a pull-back pass (tb_mech_split.hip).
template <int EN = -1, int HA = -1>
TB_HD HD energy_pair_dir(const EnergyParams &m, const double *F9, const double (&da)[9], const double (&db)[9], const double (&f0)[3], const double (&s0)[3],
                         const double (&n0)[3], double Ta)
{
    HD F[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) F[i][j] = HD{F9[3 * i + j], da[3 * i + j], db[3 * i + j], 0.0};
    if constexpr (EN >= 0) return material_psi<HD, EN, HA>(m, F, f0, s0, n0, Ta);
    if (m.prestressed) {
        HD Fe[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) Fe[i][j] = F[i][0] * m.G[j] + F[i][1] * m.G[3 + j] + F[i][2] * m.G[6 + j];
        return material_psi<HD>(m, Fe, f0, s0, n0, Ta);
    }
    return material_psi<HD>(m, F, f0, s0, n0, Ta);
}

} // namespace tbk
