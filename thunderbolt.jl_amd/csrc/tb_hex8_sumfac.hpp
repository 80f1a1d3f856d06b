// tb_hex8_sumfac.hpp — mass and diffusion element matrices of the trilinear hexahedron with the 2×2×2 Gauss rule,
// evaluated by sum factorisation.
//
// Same sums as src/modeling/core/mass.jl:28-43 (Mₑ[i,j] += ρ NᵢNⱼ dΩ) and src/modeling/core/diffusion.jl:28-50
// (Kₑ[i,j] −= ∇Nⱼ·D·∇Nᵢ dΩ) with the geometry stage of src/ferrite-addons/PR883.jl:253-263,280-291,367-387 — only the order
// of the additions differs (rounding-level, ≲1e-15 relative).  Why: the straight 8-point loop costs ≈320 FP64 instructions
// per point (≈2 560 per cell for Kₑ, ≈850 more for Mₑ with its own geometry pass) and the patch kernels were bound by
// FP64 issue; here a cell costs ≈1 400 for both matrices together.
//
//  * Nₐ(ξ) = Π_d φ(aᵈ, ξᵈ), φ ∈ {p, m} = ½(1 ± g) at the Gauss coordinates ±g; ∂Nₐ/∂ξᵈ = ±½ · (the other two factors).
//  * Kₑ[i,j] = Σ_q Σ_mn G_q[m][n] ∂ₘNᵢ ∂ₙNⱼ with G_q = −¼·(1/detJ)·A·D·Aᵀ, A = adj J (the ±½·±½ of the derivative factors is
//    the ¼): the Gauss sum factors per direction, so the 48 values G_q[mn] contract to 27 "diagonal" values Y_dd[t][t′] and
//    36 "cross" values X_de[a][b][t] (t ∈ {00,01,11}: pair type of two node bits, a, b: node bits) in ≈270 operations, and
//    every Kₑ entry is a signed sum of 9 of them.
//  * Mₑ[i,j] = Z[t₁][t₂][t₃], 27 distinct values, ≈114 operations from the 8 values ρ_q·detJ_q.
//  * J by columns: ∂x/∂ξ depends on (η, ζ) only — 4 distinct columns instead of 8 — and so on (72 FMAs per cell).
#pragma once
#include "tb_elem.hpp"

namespace tbk {

struct SF {
    static constexpr double g = 0.5773502691896258;
    static constexpr double p = 0.5 * (1.0 + g), m = 0.5 * (1.0 - g);
    TB_HD static constexpr int bit(int a, int d) { return Hex8<2>::sgn(a, d) > 0 ? 1 : 0; }
    TB_HD static constexpr double f(int b, int qb) { return b == qb ? p : m; }                  // φ(node bit, Gauss bit)
    TB_HD static constexpr double w(int t, int qb) { return t == 1 ? p * m : (t == 0 ? f(0, qb) * f(0, qb) : f(1, qb) * f(1, qb)); }
    TB_HD static constexpr int ty(int i, int j, int d) { return bit(i, d) + bit(j, d); }        // pair type of nodes i, j in direction d
    TB_HD static constexpr int sg(int i, int j, int d, int e) { return Hex8<2>::sgn(i, d) * Hex8<2>::sgn(j, e); }
};

__device__ __forceinline__ double sf_rcp(double y)
{
    double r = __builtin_amdgcn_rcp(y);
    r = fma(fma(-y, r, 1.0), r, r);
    r = fma(fma(-y, r, 1.0), r, r);
    return r;
}

// Geometry of the 8 Gauss points: G[q][xx,xy,xz,yy,yz,zz] = −¼/detJ · A·D·Aᵀ (only when WANT_K) and dq[q] = detJ_q.
// Dq(q, D6) supplies the symmetric tensor (xx,xy,xz,yy,yz,zz) of point q.  Returns false when some detJ ≤ 0.
// FENCE: keep the points' instruction streams apart (scheduling barrier after each point) — for per-point tensor loads, which
// the scheduler would otherwise hoist all at once (48 doubles in flight → spills).
// DIAG: the tensor is diagonal (D[1] = D[2] = D[4] = 0: isotropic or axis-aligned conductivities) — H = A·D is 9 products instead of 27.
// ISO: the coordinates x were mapped by L⁻¹ of the Cholesky factor D = L·Lᵀ of a constant positive definite tensor before the call: in those
// coordinates the tensor is the identity (anisotropic diffusion = isotropic diffusion in stretched coordinates), J = L·J′ gives
// A·D·Aᵀ = det L²·A′A′ᵀ and det J = det L·det J′, so G = −¼·det L/det J′ · A′A′ᵀ with no tensor product at all; `sfac` = −¼·det L (−¼ otherwise),
// dq returns det J′ (the caller folds det L into the density).
template <bool WANT_K, bool FENCE, bool DIAG = false, bool ISO = false, class DFn>
__device__ __forceinline__ bool hex8_sf_geometry(const double (&x)[8][3], DFn &&Dq, double (&G)[8][6], double (&dq)[8], double sfac = -0.25)
{
    GeoCoeffs<Hex8<2>> gc;
    geo_prepare(x, gc);
    constexpr double gg = SF::g;
    // columns of J: col0[q2 + 2 q3], col1[q1 + 2 q3], col2[q1 + 2 q2]
    double c0[4][3], c1[4][3], c2[4][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int b = 0; b < 2; ++b) { // b: bit of the slow coordinate of the pair
            const double sb = b ? gg : -gg;
            const double a0 = fma(gc.c[6][i], sb, gc.c[1][i]), a1 = fma(gc.c[7][i], sb, gc.c[4][i]); // ∂x/∂ξ = (c1 + c6 ζ) + η (c4 + c7 ζ)
            c0[0 + 2 * b][i] = fma(a1, -gg, a0); c0[1 + 2 * b][i] = fma(a1, gg, a0);
            const double b0 = fma(gc.c[5][i], sb, gc.c[2][i]);                                         // ∂x/∂η = (c2 + c5 ζ) + ξ (c4 + c7 ζ)
            c1[0 + 2 * b][i] = fma(a1, -gg, b0); c1[1 + 2 * b][i] = fma(a1, gg, b0);
            const double d0 = fma(gc.c[5][i], sb, gc.c[3][i]), d1 = fma(gc.c[7][i], sb, gc.c[6][i]); // ∂x/∂ζ = (c3 + c5 η) + ξ (c6 + c7 η)
            c2[0 + 2 * b][i] = fma(d1, -gg, d0); c2[1 + 2 * b][i] = fma(d1, gg, d0);
        }
    }
    bool ok = true;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int q1 = q & 1, q2 = (q >> 1) & 1, q3 = q >> 2;
        const double(&u)[3] = c0[q2 + 2 * q3];
        const double(&v)[3] = c1[q1 + 2 * q3];
        const double(&z)[3] = c2[q1 + 2 * q2];
        // J[i][0] = u[i], J[i][1] = v[i], J[i][2] = z[i];  A = adj J (rows as in geometry_rt's J⁻¹·det)
        double A[3][3];
        A[0][0] = v[1] * z[2] - z[1] * v[2]; A[0][1] = z[0] * v[2] - v[0] * z[2]; A[0][2] = v[0] * z[1] - z[0] * v[1];
        A[1][0] = z[1] * u[2] - u[1] * z[2]; A[1][1] = u[0] * z[2] - z[0] * u[2]; A[1][2] = z[0] * u[1] - u[0] * z[1];
        A[2][0] = u[1] * v[2] - v[1] * u[2]; A[2][1] = v[0] * u[2] - u[0] * v[2]; A[2][2] = u[0] * v[1] - v[0] * u[1];
        const double det = u[0] * A[0][0] + u[1] * A[0][1] + u[2] * A[0][2];
        ok = ok && (det > 0.0);
        dq[q] = det;
        if constexpr (WANT_K && ISO) {
            const double s = sfac * sf_rcp(det);
            int k = 0;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = a; b < 3; ++b) G[q][k++] = s * (A[a][0] * A[b][0] + A[a][1] * A[b][1] + A[a][2] * A[b][2]);
        } else if constexpr (WANT_K) {
            double D[6];
            Dq(q, D);
            const double s = sfac * sf_rcp(det);
            double H[3][3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                if constexpr (DIAG) {
                    H[a][0] = A[a][0] * D[0]; H[a][1] = A[a][1] * D[3]; H[a][2] = A[a][2] * D[5];
                } else {
                    H[a][0] = A[a][0] * D[0] + A[a][1] * D[1] + A[a][2] * D[2];
                    H[a][1] = A[a][0] * D[1] + A[a][1] * D[3] + A[a][2] * D[4];
                    H[a][2] = A[a][0] * D[2] + A[a][1] * D[4] + A[a][2] * D[5];
                }
            }
            int k = 0;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = a; b < 3; ++b) G[q][k++] = s * (H[a][0] * A[b][0] + H[a][1] * A[b][1] + H[a][2] * A[b][2]);
        }
        if constexpr (FENCE) __builtin_amdgcn_sched_barrier(0);
    }
    return ok;
}

// Contraction of one symmetric component over the Gauss points.
// diag: Y[te][tf] = Σ_q G[q] w(te, q_e) w(tf, q_f), the sum over q_d taken first (e < f the two other directions)
template <int COMP, int DIR>
__device__ __forceinline__ void hex8_sf_diag(const double (&G)[8][6], double (&Y)[3][3])
{
    constexpr int E = DIR == 0 ? 1 : 0, F = DIR == 2 ? 1 : 2;
    double S[2][2]; // [q_f][q_e]
#pragma unroll
    for (int qf = 0; qf < 2; ++qf)
#pragma unroll
        for (int qe = 0; qe < 2; ++qe) S[qf][qe] = G[(qe << E) | (qf << F)][COMP] + G[(qe << E) | (qf << F) | (1 << DIR)][COMP];
    double T[3][2]; // [te][q_f]
#pragma unroll
    for (int qf = 0; qf < 2; ++qf) {
        T[0][qf] = fma(SF::w(0, 1), S[qf][1], SF::w(0, 0) * S[qf][0]);
        T[1][qf] = SF::w(1, 0) * (S[qf][0] + S[qf][1]);
        T[2][qf] = fma(SF::w(2, 1), S[qf][1], SF::w(2, 0) * S[qf][0]);
    }
#pragma unroll
    for (int te = 0; te < 3; ++te) {
        Y[te][0] = fma(SF::w(0, 1), T[te][1], SF::w(0, 0) * T[te][0]);
        Y[te][1] = SF::w(1, 0) * (T[te][0] + T[te][1]);
        Y[te][2] = fma(SF::w(2, 1), T[te][1], SF::w(2, 0) * T[te][0]);
    }
}
// cross: X[a][b][t] = Σ_q G[q] φ(a, q_d) φ(b, q_e) w(t, q_f)   (d < e, f the remaining direction)
template <int COMP, int D, int E>
__device__ __forceinline__ void hex8_sf_cross(const double (&G)[8][6], double (&X)[2][2][3])
{
    constexpr int F = 3 - D - E;
    double A1[2][2][2]; // [a][q_e][q_f]
#pragma unroll
    for (int qe = 0; qe < 2; ++qe)
#pragma unroll
        for (int qf = 0; qf < 2; ++qf) {
            const double g0 = G[(qe << E) | (qf << F)][COMP], g1 = G[(qe << E) | (qf << F) | (1 << D)][COMP];
            A1[0][qe][qf] = fma(SF::f(0, 1), g1, SF::f(0, 0) * g0);
            A1[1][qe][qf] = fma(SF::f(1, 1), g1, SF::f(1, 0) * g0);
        }
    double B1[2][2][2]; // [a][b][q_f]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int qf = 0; qf < 2; ++qf) {
            B1[a][0][qf] = fma(SF::f(0, 1), A1[a][1][qf], SF::f(0, 0) * A1[a][0][qf]);
            B1[a][1][qf] = fma(SF::f(1, 1), A1[a][1][qf], SF::f(1, 0) * A1[a][0][qf]);
        }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            X[a][b][0] = fma(SF::w(0, 1), B1[a][b][1], SF::w(0, 0) * B1[a][b][0]);
            X[a][b][1] = SF::w(1, 0) * (B1[a][b][0] + B1[a][b][1]);
            X[a][b][2] = fma(SF::w(2, 1), B1[a][b][1], SF::w(2, 0) * B1[a][b][0]);
        }
}

// Linear form bₑ[j] = Σ_q f(x_q) Nⱼ(ξ_q) detJ_q (src/modeling/core/analytical_coefficient.jl:89-99) by the same factorisation: Jacobian columns and
// Gauss-point positions from the modal geometry (4 and 8 distinct values per component instead of 8 × 8 products), the 8 weighted values
// f·detJ contracted to the 8 nodes direction by direction.  F(q, xq) evaluates the source at Gauss point q.  ≈ 330 FP64 instructions per cell
// plus 8 source evaluations (the 8-point loop of tb_assembly.hip: ≈ 900).  Returns false when some detJ ≤ 0.
template <class FFn>
__device__ __forceinline__ bool hex8_sf_source(const double (&x)[8][3], FFn &&F, double (&be)[8])
{
    GeoCoeffs<Hex8<2>> gc;
    geo_prepare(x, gc);
    constexpr double gg = SF::g;
    double c0[4][3], c1[4][3], c2[4][3], xq[8][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const double sb = b ? gg : -gg;
            const double a0 = fma(gc.c[6][i], sb, gc.c[1][i]), a1 = fma(gc.c[7][i], sb, gc.c[4][i]);
            c0[0 + 2 * b][i] = fma(a1, -gg, a0); c0[1 + 2 * b][i] = fma(a1, gg, a0);
            const double b0 = fma(gc.c[5][i], sb, gc.c[2][i]);
            c1[0 + 2 * b][i] = fma(a1, -gg, b0); c1[1 + 2 * b][i] = fma(a1, gg, b0);
            const double d0 = fma(gc.c[5][i], sb, gc.c[3][i]), d1 = fma(gc.c[7][i], sb, gc.c[6][i]);
            c2[0 + 2 * b][i] = fma(d1, -gg, d0); c2[1 + 2 * b][i] = fma(d1, gg, d0);
            // position: x = (c0 + c3 ζ) + η (c2 + c5 ζ) + ξ [(c1 + c6 ζ) + η (c4 + c7 ζ)], b = bit of ζ
            const double p0 = fma(gc.c[3][i], sb, gc.c[0][i]);
#pragma unroll
            for (int e = 0; e < 2; ++e) { // bit of η
                const double se = e ? gg : -gg;
                const double r0 = fma(b0, se, p0), r1 = fma(a1, se, a0);
                xq[0 + 2 * e + 4 * b][i] = fma(r1, -gg, r0);
                xq[1 + 2 * e + 4 * b][i] = fma(r1, gg, r0);
            }
        }
    }
    bool ok = true;
    double fw[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int q1 = q & 1, q2 = (q >> 1) & 1, q3 = q >> 2;
        const double(&u)[3] = c0[q2 + 2 * q3];
        const double(&v)[3] = c1[q1 + 2 * q3];
        const double(&z)[3] = c2[q1 + 2 * q2];
        const double det = u[0] * (v[1] * z[2] - z[1] * v[2]) + u[1] * (z[0] * v[2] - v[0] * z[2]) + u[2] * (v[0] * z[1] - z[0] * v[1]);
        ok = ok && (det > 0.0);
        fw[q] = F(q, xq[q]) * det;
    }
    double A1[2][2][2]; // [b1][q2][q3]
#pragma unroll
    for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
        for (int q3 = 0; q3 < 2; ++q3) {
            const double f0 = fw[2 * q2 + 4 * q3], f1 = fw[1 + 2 * q2 + 4 * q3];
            A1[0][q2][q3] = fma(SF::f(0, 1), f1, SF::f(0, 0) * f0);
            A1[1][q2][q3] = fma(SF::f(1, 1), f1, SF::f(1, 0) * f0);
        }
    double B1[2][2][2]; // [b1][b2][q3]
#pragma unroll
    for (int b1 = 0; b1 < 2; ++b1)
#pragma unroll
        for (int q3 = 0; q3 < 2; ++q3) {
            B1[b1][0][q3] = fma(SF::f(0, 1), A1[b1][1][q3], SF::f(0, 0) * A1[b1][0][q3]);
            B1[b1][1][q3] = fma(SF::f(1, 1), A1[b1][1][q3], SF::f(1, 0) * A1[b1][0][q3]);
        }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int b1 = SF::bit(j, 0), b2 = SF::bit(j, 1), b3 = SF::bit(j, 2);
        be[j] = fma(SF::f(b3, 1), B1[b1][b2][1], SF::f(b3, 0) * B1[b1][b2][0]);
    }
    return ok;
}

struct Hex8SFK {
    double Y11[3][3], Y22[3][3], Y33[3][3]; // [type lower dir][type higher dir] of the two remaining directions
    double X12[2][2][3], X13[2][2][3], X23[2][2][3];
};

__device__ __forceinline__ void hex8_sf_contract(const double (&G)[8][6], Hex8SFK &c)
{
    hex8_sf_diag<0, 0>(G, c.Y11);
    hex8_sf_diag<3, 1>(G, c.Y22);
    hex8_sf_diag<5, 2>(G, c.Y33);
    hex8_sf_cross<1, 0, 1>(G, c.X12);
    hex8_sf_cross<2, 0, 2>(G, c.X13);
    hex8_sf_cross<4, 1, 2>(G, c.X23);
}

// Kₑ[i][j] from the contracted values (I, J compile-time after unrolling): a signed sum of nine of them, as a chain of eight additions.  The tree form
// ((t₀ + t₁) + (t₂ + t₃)) + x₉ (−DTB_SF_ENTRY_TREE) halves the dependent depth and was measured at 216³: 1.705 / 1.702 ms against 1.704 / 1.693 ms for the
// chain — two waves per SIMD already cover the FP64 latency — and it costs a register the fused kernel does not have (8 B of scratch): not used.
__device__ __forceinline__ double hex8_sf_pair(bool pa, double a, bool pb, double b)
{
    return pa ? (pb ? a + b : a - b) : (pb ? b - a : -(a + b));
}
__device__ __forceinline__ double hex8_sf_entry(const Hex8SFK &c, int i, int j)
{
    const int t0 = SF::ty(i, j, 0), t1 = SF::ty(i, j, 1), t2 = SF::ty(i, j, 2);
    const int bi0 = SF::bit(i, 0), bi1 = SF::bit(i, 1), bi2 = SF::bit(i, 2);
    const int bj0 = SF::bit(j, 0), bj1 = SF::bit(j, 1), bj2 = SF::bit(j, 2);
#ifndef TB_SF_ENTRY_TREE
    double v = SF::sg(i, j, 0, 0) > 0 ? c.Y11[t1][t2] : -c.Y11[t1][t2];
    v += SF::sg(i, j, 1, 1) > 0 ? c.Y22[t0][t2] : -c.Y22[t0][t2];
    v += SF::sg(i, j, 2, 2) > 0 ? c.Y33[t0][t1] : -c.Y33[t0][t1];
    v += SF::sg(i, j, 0, 1) > 0 ? c.X12[bj0][bi1][t2] : -c.X12[bj0][bi1][t2];
    v += SF::sg(j, i, 0, 1) > 0 ? c.X12[bi0][bj1][t2] : -c.X12[bi0][bj1][t2];
    v += SF::sg(i, j, 0, 2) > 0 ? c.X13[bj0][bi2][t1] : -c.X13[bj0][bi2][t1];
    v += SF::sg(j, i, 0, 2) > 0 ? c.X13[bi0][bj2][t1] : -c.X13[bi0][bj2][t1];
    v += SF::sg(i, j, 1, 2) > 0 ? c.X23[bj1][bi2][t0] : -c.X23[bj1][bi2][t0];
    v += SF::sg(j, i, 1, 2) > 0 ? c.X23[bi1][bj2][t0] : -c.X23[bi1][bj2][t0];
    return v;
#else
    const double a0 = hex8_sf_pair(SF::sg(i, j, 0, 0) > 0, c.Y11[t1][t2], SF::sg(i, j, 1, 1) > 0, c.Y22[t0][t2]);
    const double a1 = hex8_sf_pair(SF::sg(i, j, 2, 2) > 0, c.Y33[t0][t1], SF::sg(i, j, 0, 1) > 0, c.X12[bj0][bi1][t2]);
    const double a2 = hex8_sf_pair(SF::sg(j, i, 0, 1) > 0, c.X12[bi0][bj1][t2], SF::sg(i, j, 0, 2) > 0, c.X13[bj0][bi2][t1]);
    const double a3 = hex8_sf_pair(SF::sg(j, i, 0, 2) > 0, c.X13[bi0][bj2][t1], SF::sg(i, j, 1, 2) > 0, c.X23[bj1][bi2][t0]);
    const double x9 = c.X23[bi1][bj2][t0];
    const double s = (a0 + a1) + (a2 + a3);
    return SF::sg(j, i, 1, 2) > 0 ? s + x9 : s - x9;
#endif
}

// Mₑ: Z[t₁][t₂][t₃] = Σ_q d_q w(t₁,q₁) w(t₂,q₂) w(t₃,q₃); Mₑ[i][j] = Z[ty(i,j,0)][ty(i,j,1)][ty(i,j,2)]
__device__ __forceinline__ void hex8_sf_mass(const double (&d)[8], double (&Z)[3][3][3])
{
    double U[3][2][2]; // [t1][q2][q3]
#pragma unroll
    for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
        for (int q3 = 0; q3 < 2; ++q3) {
            const double d0 = d[2 * q2 + 4 * q3], d1 = d[1 + 2 * q2 + 4 * q3];
            U[0][q2][q3] = fma(SF::w(0, 1), d1, SF::w(0, 0) * d0);
            U[1][q2][q3] = SF::w(1, 0) * (d0 + d1);
            U[2][q2][q3] = fma(SF::w(2, 1), d1, SF::w(2, 0) * d0);
        }
    double V[3][3][2]; // [t1][t2][q3]
#pragma unroll
    for (int t1 = 0; t1 < 3; ++t1)
#pragma unroll
        for (int q3 = 0; q3 < 2; ++q3) {
            V[t1][0][q3] = fma(SF::w(0, 1), U[t1][1][q3], SF::w(0, 0) * U[t1][0][q3]);
            V[t1][1][q3] = SF::w(1, 0) * (U[t1][0][q3] + U[t1][1][q3]);
            V[t1][2][q3] = fma(SF::w(2, 1), U[t1][1][q3], SF::w(2, 0) * U[t1][0][q3]);
        }
#pragma unroll
    for (int t1 = 0; t1 < 3; ++t1)
#pragma unroll
        for (int t2 = 0; t2 < 3; ++t2) {
            Z[t1][t2][0] = fma(SF::w(0, 1), V[t1][t2][1], SF::w(0, 0) * V[t1][t2][0]);
            Z[t1][t2][1] = SF::w(1, 0) * (V[t1][t2][0] + V[t1][t2][1]);
            Z[t1][t2][2] = fma(SF::w(2, 1), V[t1][t2][1], SF::w(2, 0) * V[t1][t2][0]);
        }
}

} // namespace tbk
