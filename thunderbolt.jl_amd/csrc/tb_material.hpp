// tb_material.hpp — hand-derived first Piola–Kirchhoff stress P = ∂Ψ/∂F and tangent 𝔸 = ∂²Ψ/∂F² of the
// Holzapfel–Ogden 2009 energy with SimpleCompressionPenalty.
//
// The reference obtains both by forward-mode AD of Ψ(F) (Tensors.gradient / Tensors.hessian,
// src/modeling/solid/materials.jl:1025-1040) with Ψ from src/modeling/solid/energies.jl:147-168 and
// U(I₃) = β(I₃ − 1 − 2 log √I₃) from :83-87.  AD types do not exist in device C++, so the derivatives are
// written out; parity with the AD result is checked against the hyper-dual oracle.
//
// Notation: F_ij = ∂x_i/∂X_j, J = det F, C = FᵀF, g = I₃^{-1/3} = J^{-2/3}, Ī₁ = g·tr C,
//   a = F f₀, b = F s₀, I₄ᶠ = a·a, I₄ˢ = b·b, I₈ = a·b.
//   ∂J/∂F_ij = J F⁻¹_ji,  ∂F⁻¹_ji/∂F_kl = −F⁻¹_jk F⁻¹_li.
//   B_ij   = ∂Ī₁/∂F_ij = g (2F_ij − ⅔ tr C · F⁻¹_ji)
//   ∂²Ī₁   = −⅔ F⁻¹_lk B_ij + g [2 δ_ik δ_jl − 4/3 F_kl F⁻¹_ji + ⅔ tr C · F⁻¹_jk F⁻¹_li]
//   Ψ₁ = a/(2b)(e^{b(Ī₁−3)} − 1):   P += ½a E₁ B,          𝔸 += ½a E₁ (b B⊗B + ∂²Ī₁)
//   Ψ₄ = aᶠ/(2bᶠ)(e^{bᶠ(I₄−1)²} − 1) if I₄ ≥ 1:  w₁ = aᶠ(I₄−1)E₄, w₂ = aᶠE₄(1 + 2bᶠ(I₄−1)²)
//        P += w₁·2 a⊗f₀,   𝔸_ijkl += 4 w₂ a_i f_j a_k f_l + 2 w₁ δ_ik f_j f_l        (same for s₀)
//   Ψ₈ = aᶠˢ/(2bᶠˢ)(e^{bᶠˢ I₈²} − 1):  M = b⊗f₀ + a⊗s₀, v₁ = aᶠˢ I₈ E₈, v₂ = aᶠˢ E₈ (1 + 2bᶠˢ I₈²)
//        P += v₁ M,         𝔸_ijkl += v₂ M_ij M_kl + v₁ δ_ik (s_j f_l + f_j s_l)
//   U  = β(J² − 1 − 2 ln J):  P += 2β(J²−1) F⁻ᵀ,  𝔸_ijkl += 4βJ² F⁻¹_ji F⁻¹_lk − 2β(J²−1) F⁻¹_jk F⁻¹_li
//   active stress (ActiveStressModel, materials.jl:1200-1266; SimpleActiveStress, active.jl:100-113):
//        Pᵃ = Ta (F f₀)⊗f₀/‖F f₀‖ = ∂(Ta √I₄ᶠ)/∂F — the Ψ₄ pattern with w₁ = Ta/(2√I₄ᶠ), w₂ = −Ta/(4 I₄ᶠ^{3/2})
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

namespace tbk {

#ifndef TB_HD
#define TB_HD __host__ __device__ __forceinline__
#endif

struct HOParams {
    double a, b, af, bf, as, bs, afs, bfs, beta; // struct field order of HolzapfelOgden2009Model, energies.jl:136-146
    double f[3], s[3], n[3];                     // microstructure frame (ConstantCoefficient(OrthotropicMicrostructure))
    double Ta;                                   // active tension Tmax·𝓝 at the point (ActiveStressModel + SimpleActiveStress); 0 = passive
    double Tb;                                   // condensed internal variable: + Tb ∂λ/∂F ⊗ ∂λ/∂F in the tangent (d𝓝/dλ incl. the local corrector)
};

// returns Ψ; P[3i+j], A[9(3i+j) + 3k+l] (A only when NEED_A)
// `scale` multiplies P and A on output (the kernels pass dΩ and point A straight at LDS)
template <bool NEED_A>
TB_HD double ho_stress_tangent(const HOParams &m, const double (&F)[3][3], double (&P)[9], double *A, double scale = 1.0)
{
    // J, F⁻¹
    const double c00 = F[1][1] * F[2][2] - F[1][2] * F[2][1];
    const double c01 = F[1][2] * F[2][0] - F[1][0] * F[2][2];
    const double c02 = F[1][0] * F[2][1] - F[1][1] * F[2][0];
    const double J = F[0][0] * c00 + F[0][1] * c01 + F[0][2] * c02;
    const double iJ = 1.0 / J;
    double Fi[3][3];
    Fi[0][0] = c00 * iJ; Fi[0][1] = (F[0][2] * F[2][1] - F[0][1] * F[2][2]) * iJ; Fi[0][2] = (F[0][1] * F[1][2] - F[0][2] * F[1][1]) * iJ;
    Fi[1][0] = c01 * iJ; Fi[1][1] = (F[0][0] * F[2][2] - F[0][2] * F[2][0]) * iJ; Fi[1][2] = (F[0][2] * F[1][0] - F[0][0] * F[1][2]) * iJ;
    Fi[2][0] = c02 * iJ; Fi[2][1] = (F[0][1] * F[2][0] - F[0][0] * F[2][1]) * iJ; Fi[2][2] = (F[0][0] * F[1][1] - F[0][1] * F[1][0]) * iJ;
    const double J2 = J * J;
    double trC = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) trC += F[i][j] * F[i][j];
    const double g = 1.0 / cbrt(J2);
    const double I1b = trC * g;
    // fibre kinematics
    double av[3], bv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        av[i] = F[i][0] * m.f[0] + F[i][1] * m.f[1] + F[i][2] * m.f[2];
        bv[i] = F[i][0] * m.s[0] + F[i][1] * m.s[1] + F[i][2] * m.s[2];
    }
    const double I4f = av[0] * av[0] + av[1] * av[1] + av[2] * av[2];
    const double I4s = bv[0] * bv[0] + bv[1] * bv[1] + bv[2] * bv[2];
    const double I8 = av[0] * bv[0] + av[1] * bv[1] + av[2] * bv[2];

    const double E1 = exp(m.b * (I1b - 3.0));
    const double E8 = exp(m.bfs * I8 * I8);
    const bool onf = I4f >= 1.0, ons = I4s >= 1.0;
    const double df = I4f - 1.0, ds = I4s - 1.0;
    const double E4f = onf ? exp(m.bf * df * df) : 1.0;
    const double E4s = ons ? exp(m.bs * ds * ds) : 1.0;
    double psi = m.a / (2.0 * m.b) * (E1 - 1.0) + m.afs / (2.0 * m.bfs) * (E8 - 1.0) + m.beta * (J2 - 1.0 - 2.0 * log(J));
    if (onf) psi += m.af / (2.0 * m.bf) * (E4f - 1.0);
    if (ons) psi += m.as / (2.0 * m.bs) * (E4s - 1.0);

    const double h1 = 0.5 * m.a * E1;
    const double lamf = sqrt(I4f);
    const double w1f = (onf ? m.af * df * E4f : 0.0) + 0.5 * m.Ta / lamf, w2f = (onf ? m.af * E4f * (1.0 + 2.0 * m.bf * df * df) : 0.0) - 0.25 * m.Ta / (lamf * I4f) + 0.25 * m.Tb / I4f;
    psi += m.Ta * lamf;
    const double w1s = ons ? m.as * ds * E4s : 0.0, w2s = ons ? m.as * E4s * (1.0 + 2.0 * m.bs * ds * ds) : 0.0;
    const double v1 = m.afs * I8 * E8, v2 = m.afs * E8 * (1.0 + 2.0 * m.bfs * I8 * I8);
    const double u1 = 2.0 * m.beta * (J2 - 1.0), u2 = 4.0 * m.beta * J2;

    double B[3][3], M[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            B[i][j] = g * (2.0 * F[i][j] - (2.0 / 3.0) * trC * Fi[j][i]);
            M[i][j] = bv[i] * m.f[j] + av[i] * m.s[j];
            P[3 * i + j] = scale * (h1 * B[i][j] + 2.0 * w1f * av[i] * m.f[j] + 2.0 * w1s * bv[i] * m.s[j] + v1 * M[i][j] + u1 * Fi[j][i]);
        }
    if constexpr (NEED_A) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int l = 0; l < 3; ++l) {
                        const double dik = i == k ? 1.0 : 0.0, djl = j == l ? 1.0 : 0.0;
                        const double d2I1 = -(2.0 / 3.0) * Fi[l][k] * B[i][j] +
                                            g * (2.0 * dik * djl - (4.0 / 3.0) * F[k][l] * Fi[j][i] + (2.0 / 3.0) * trC * Fi[j][k] * Fi[l][i]);
                        double t = h1 * (m.b * B[i][j] * B[k][l] + d2I1);
                        t += 4.0 * w2f * av[i] * m.f[j] * av[k] * m.f[l] + 2.0 * w1f * dik * m.f[j] * m.f[l];
                        t += 4.0 * w2s * bv[i] * m.s[j] * bv[k] * m.s[l] + 2.0 * w1s * dik * m.s[j] * m.s[l];
                        t += v2 * M[i][j] * M[k][l] + v1 * dik * (m.s[j] * m.f[l] + m.f[j] * m.s[l]);
                        t += u2 * Fi[j][i] * Fi[l][k] - u1 * Fi[j][k] * Fi[l][i];
                        A[9 * (3 * i + j) + 3 * k + l] = scale * t;
                    }
    }
    return psi;
}

} // namespace tbk

// ---- split evaluation for the kernels: per-point common block, then one (i,j) row of P / 𝔸 per lane ----
// ho_stress_tangent keeps a whole 9×9 tangent in registers on ONE lane per Gauss point; inside a workgroup that is
// the critical path.  The kernels instead compute the ~46 shared quantities once per point (ho_common → LDS) and let
// 9 lanes per point build one row (i,j) of 𝔸 each (ho_row): same arithmetic, 4× shorter critical path.
namespace tbk {

enum { HOC_FI = 0, HOC_B = 9, HOC_M = 18, HOC_AV = 27, HOC_BV = 30, HOC_H1 = 33, HOC_H1B, HOC_W1F2, HOC_W2F4, HOC_W1S2, HOC_W2S4,
       HOC_V1, HOC_V2, HOC_U1, HOC_U2, HOC_G, HOC_TRC, HOC_PSI, HOC_FV, HOC_SV = HOC_FV + 3, HOC_SIZE = 52 };

// PSI = false (the assembly kernels, which never read HOC_PSI): the energy — a logarithm and four quotients of material constants — is not evaluated
template <bool PSI = true>
TB_HD void ho_common(const HOParams &m, const double (&F)[3][3], double *C)
{
    const double c00 = F[1][1] * F[2][2] - F[1][2] * F[2][1];
    const double c01 = F[1][2] * F[2][0] - F[1][0] * F[2][2];
    const double c02 = F[1][0] * F[2][1] - F[1][1] * F[2][0];
    const double J = F[0][0] * c00 + F[0][1] * c01 + F[0][2] * c02;
    const double iJ = 1.0 / J;
    double Fi[3][3];
    Fi[0][0] = c00 * iJ; Fi[0][1] = (F[0][2] * F[2][1] - F[0][1] * F[2][2]) * iJ; Fi[0][2] = (F[0][1] * F[1][2] - F[0][2] * F[1][1]) * iJ;
    Fi[1][0] = c01 * iJ; Fi[1][1] = (F[0][0] * F[2][2] - F[0][2] * F[2][0]) * iJ; Fi[1][2] = (F[0][2] * F[1][0] - F[0][0] * F[1][2]) * iJ;
    Fi[2][0] = c02 * iJ; Fi[2][1] = (F[0][1] * F[2][0] - F[0][0] * F[2][1]) * iJ; Fi[2][2] = (F[0][0] * F[1][1] - F[0][1] * F[1][0]) * iJ;
    const double J2 = J * J;
    double trC = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) trC += F[i][j] * F[i][j];
    const double g = 1.0 / cbrt(J2);
    double av[3], bv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        av[i] = F[i][0] * m.f[0] + F[i][1] * m.f[1] + F[i][2] * m.f[2];
        bv[i] = F[i][0] * m.s[0] + F[i][1] * m.s[1] + F[i][2] * m.s[2];
    }
    const double I4f = av[0] * av[0] + av[1] * av[1] + av[2] * av[2];
    const double I4s = bv[0] * bv[0] + bv[1] * bv[1] + bv[2] * bv[2];
    const double I8 = av[0] * bv[0] + av[1] * bv[1] + av[2] * bv[2];
    const double E1 = exp(m.b * (trC * g - 3.0));
    const double E8 = exp(m.bfs * I8 * I8);
    const bool onf = I4f >= 1.0, ons = I4s >= 1.0;
    const double df = I4f - 1.0, ds = I4s - 1.0;
    const double E4f = onf ? exp(m.bf * df * df) : 1.0;
    const double E4s = ons ? exp(m.bs * ds * ds) : 1.0;
    double psi = 0.0;
    if constexpr (PSI) {
        psi = m.a / (2.0 * m.b) * (E1 - 1.0) + m.afs / (2.0 * m.bfs) * (E8 - 1.0) + m.beta * (J2 - 1.0 - 2.0 * log(J));
        if (onf) psi += m.af / (2.0 * m.bf) * (E4f - 1.0);
        if (ons) psi += m.as / (2.0 * m.bs) * (E4s - 1.0);
    }
    const double h1 = 0.5 * m.a * E1;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            C[HOC_FI + 3 * i + j] = Fi[i][j];
            C[HOC_B + 3 * i + j] = g * (2.0 * F[i][j] - (2.0 / 3.0) * trC * Fi[j][i]);
            C[HOC_M + 3 * i + j] = bv[i] * m.f[j] + av[i] * m.s[j];
        }
#pragma unroll
    for (int i = 0; i < 3; ++i) { C[HOC_AV + i] = av[i]; C[HOC_BV + i] = bv[i]; }
    C[HOC_H1] = h1; C[HOC_H1B] = h1 * m.b;
    const double lamf = sqrt(I4f);
    if constexpr (PSI) psi += m.Ta * lamf;
    C[HOC_W1F2] = (onf ? 2.0 * m.af * df * E4f : 0.0) + m.Ta / lamf;
    C[HOC_W2F4] = (onf ? 4.0 * m.af * E4f * (1.0 + 2.0 * m.bf * df * df) : 0.0) - m.Ta / (lamf * I4f) + m.Tb / I4f;
    C[HOC_W1S2] = ons ? 2.0 * m.as * ds * E4s : 0.0; C[HOC_W2S4] = ons ? 4.0 * m.as * E4s * (1.0 + 2.0 * m.bs * ds * ds) : 0.0;
    C[HOC_V1] = m.afs * I8 * E8; C[HOC_V2] = m.afs * E8 * (1.0 + 2.0 * m.bfs * I8 * I8);
    C[HOC_U1] = 2.0 * m.beta * (J2 - 1.0); C[HOC_U2] = 4.0 * m.beta * J2;
    C[HOC_G] = g; C[HOC_TRC] = trC; C[HOC_PSI] = psi;
#pragma unroll
    for (int i = 0; i < 3; ++i) { C[HOC_FV + i] = m.f[i]; C[HOC_SV + i] = m.s[i]; }
}

// The common block in four independent parts, one per wave of a 256-thread workgroup (lane = point): ho_common on one lane per point is the longest
// serial stretch of the mechanics kernels (a reciprocal cube root, four exponentials, a square root and five quotients behind one another, 3.2 µs of a
// 19 µs cell), while three of the workgroup's four waves wait.  Same expressions as ho_common, entry by entry (bit-identical block).
//   part 0: F⁻¹, B, g, tr C, the volumetric and isotropic factors      part 1: fibre terms, a, b, M, the frame
//   part 2: sheet terms                                                  part 3: fibre–sheet coupling
template <int PART>
TB_HD void ho_common_part(const HOParams &m, const double (&F)[3][3], double *C)
{
    if constexpr (PART == 0) {
        const double c00 = F[1][1] * F[2][2] - F[1][2] * F[2][1];
        const double c01 = F[1][2] * F[2][0] - F[1][0] * F[2][2];
        const double c02 = F[1][0] * F[2][1] - F[1][1] * F[2][0];
        const double J = F[0][0] * c00 + F[0][1] * c01 + F[0][2] * c02;
        const double iJ = 1.0 / J;
        double Fi[3][3];
        Fi[0][0] = c00 * iJ; Fi[0][1] = (F[0][2] * F[2][1] - F[0][1] * F[2][2]) * iJ; Fi[0][2] = (F[0][1] * F[1][2] - F[0][2] * F[1][1]) * iJ;
        Fi[1][0] = c01 * iJ; Fi[1][1] = (F[0][0] * F[2][2] - F[0][2] * F[2][0]) * iJ; Fi[1][2] = (F[0][2] * F[1][0] - F[0][0] * F[1][2]) * iJ;
        Fi[2][0] = c02 * iJ; Fi[2][1] = (F[0][1] * F[2][0] - F[0][0] * F[2][1]) * iJ; Fi[2][2] = (F[0][0] * F[1][1] - F[0][1] * F[1][0]) * iJ;
        const double J2 = J * J;
        double trC = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) trC += F[i][j] * F[i][j];
        const double g = 1.0 / cbrt(J2);
        const double E1 = exp(m.b * (trC * g - 3.0));
        const double h1 = 0.5 * m.a * E1;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                C[HOC_FI + 3 * i + j] = Fi[i][j];
                C[HOC_B + 3 * i + j] = g * (2.0 * F[i][j] - (2.0 / 3.0) * trC * Fi[j][i]);
            }
        C[HOC_H1] = h1; C[HOC_H1B] = h1 * m.b;
        C[HOC_U1] = 2.0 * m.beta * (J2 - 1.0); C[HOC_U2] = 4.0 * m.beta * J2;
        C[HOC_G] = g; C[HOC_TRC] = trC; C[HOC_PSI] = 0.0;
    } else {
        double av[3], bv[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            av[i] = F[i][0] * m.f[0] + F[i][1] * m.f[1] + F[i][2] * m.f[2];
            bv[i] = F[i][0] * m.s[0] + F[i][1] * m.s[1] + F[i][2] * m.s[2];
        }
        if constexpr (PART == 1) {
            const double I4f = av[0] * av[0] + av[1] * av[1] + av[2] * av[2];
            const bool onf = I4f >= 1.0;
            const double df = I4f - 1.0;
            const double E4f = onf ? exp(m.bf * df * df) : 1.0;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) C[HOC_M + 3 * i + j] = bv[i] * m.f[j] + av[i] * m.s[j];
#pragma unroll
            for (int i = 0; i < 3; ++i) { C[HOC_AV + i] = av[i]; C[HOC_BV + i] = bv[i]; }
            const double lamf = sqrt(I4f);
            C[HOC_W1F2] = (onf ? 2.0 * m.af * df * E4f : 0.0) + m.Ta / lamf;
            C[HOC_W2F4] = (onf ? 4.0 * m.af * E4f * (1.0 + 2.0 * m.bf * df * df) : 0.0) - m.Ta / (lamf * I4f) + m.Tb / I4f;
#pragma unroll
            for (int i = 0; i < 3; ++i) { C[HOC_FV + i] = m.f[i]; C[HOC_SV + i] = m.s[i]; }
        } else if constexpr (PART == 2) {
            const double I4s = bv[0] * bv[0] + bv[1] * bv[1] + bv[2] * bv[2];
            const bool ons = I4s >= 1.0;
            const double ds = I4s - 1.0;
            const double E4s = ons ? exp(m.bs * ds * ds) : 1.0;
            C[HOC_W1S2] = ons ? 2.0 * m.as * ds * E4s : 0.0; C[HOC_W2S4] = ons ? 4.0 * m.as * E4s * (1.0 + 2.0 * m.bs * ds * ds) : 0.0;
        } else {
            const double I8 = av[0] * bv[0] + av[1] * bv[1] + av[2] * bv[2];
            const double E8 = exp(m.bfs * I8 * I8);
            C[HOC_V1] = m.afs * I8 * E8; C[HOC_V2] = m.afs * E8 * (1.0 + 2.0 * m.bfs * I8 * I8);
        }
    }
}

// orthogonalize_system(f,s,n): normalise, then Gram–Schmidt without renormalising w₂ (src/utils.jl:131-139)
TB_HD void ho_orthonormal_frame(double (&f)[3], double (&s)[3], double (&n)[3])
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

// row (i,j) of P and 𝔸 from the common block (C and F may live in LDS; i, j may differ per lane)
template <bool NEED_A>
TB_HD void ho_row(const HOParams &m, const double *C, const double *F, int i, int j, double scale, double &P_ij, double *Arow)
{
    const double Bij = C[HOC_B + 3 * i + j], Mij = C[HOC_M + 3 * i + j], Fiji = C[HOC_FI + 3 * j + i];
    const double avi = C[HOC_AV + i], bvi = C[HOC_BV + i];
    const double fj = C[HOC_FV + j], sj = C[HOC_SV + j]; // frame of THIS point (constant or interpolated nodal field)
    const double h1 = C[HOC_H1], h1b = C[HOC_H1B], w1f2 = C[HOC_W1F2], w2f4 = C[HOC_W2F4], w1s2 = C[HOC_W1S2], w2s4 = C[HOC_W2S4];
    const double v1 = C[HOC_V1], v2 = C[HOC_V2], u1 = C[HOC_U1], u2 = C[HOC_U2], g = C[HOC_G], trC = C[HOC_TRC];
    P_ij = scale * (h1 * Bij + w1f2 * avi * fj + w1s2 * bvi * sj + v1 * Mij + u1 * Fiji);
    if constexpr (NEED_A) {
        const double Fij_[3] = {C[HOC_FI + 3 * j + 0], C[HOC_FI + 3 * j + 1], C[HOC_FI + 3 * j + 2]}; // F⁻¹[j][k]
        const double Fi_i[3] = {C[HOC_FI + 0 + i], C[HOC_FI + 3 + i], C[HOC_FI + 6 + i]};             // F⁻¹[l][i]
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                const double dik = i == k ? 1.0 : 0.0, djl = j == l ? 1.0 : 0.0;
                const double Filk = C[HOC_FI + 3 * l + k], Bkl = C[HOC_B + 3 * k + l], Mkl = C[HOC_M + 3 * k + l];
                const double d2I1 = -(2.0 / 3.0) * Filk * Bij + g * (2.0 * dik * djl - (4.0 / 3.0) * F[3 * k + l] * Fiji + (2.0 / 3.0) * trC * Fij_[k] * Fi_i[l]);
                double t = h1 * d2I1 + h1b * Bij * Bkl;
                const double fl = C[HOC_FV + l], sl = C[HOC_SV + l];
                t += w2f4 * avi * fj * C[HOC_AV + k] * fl + w1f2 * dik * fj * fl;
                t += w2s4 * bvi * sj * C[HOC_BV + k] * sl + w1s2 * dik * sj * sl;
                t += v2 * Mij * Mkl + v1 * dik * (sj * fl + fj * sl);
                t += u2 * Fiji * Filk - u1 * Fij_[k] * Fi_i[l];
                Arow[3 * k + l] = scale * t;
            }
    }
}

} // namespace tbk
