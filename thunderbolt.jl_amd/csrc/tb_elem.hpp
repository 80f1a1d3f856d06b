// tb_elem.hpp — compile-time reference-element tables and the per-quadrature-point geometry stage.
//
// Everything that depends only on the reference element (N, dN/dξ, Gauss points, weights) is a
// constexpr function of literal indices; the kernels unroll all q / a / i / j loops, so these fold
// into instruction immediates — no table loads, no LDS staging needed for Q1.
// Arithmetic restated: src/ferrite-addons/PR883.jl:253-263 (J = Σ xₐ ⊗ dMₐ/dξ), :280-291
// (dN/dx = dN/dξ · J⁻¹), :367-387 (detJ·w, detJ > 0 check).  Reference-element conventions are
// Ferrite.jl's (third party): see DESIGN.md §conventions.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

namespace tbk {

#define TB_HD __host__ __device__ __forceinline__

template <int ORDER> struct Gauss;
template <> struct Gauss<1> {
    TB_HD static constexpr double x(int) { return 0.0; }
    TB_HD static constexpr double w(int) { return 2.0; }
};
template <> struct Gauss<2> {
    TB_HD static constexpr double x(int i) { return i == 0 ? -0.5773502691896258 : 0.5773502691896258; }
    TB_HD static constexpr double w(int) { return 1.0; }
};
template <> struct Gauss<3> {
    TB_HD static constexpr double x(int i) { return i == 0 ? -0.7745966692414834 : i == 1 ? 0.0 : 0.7745966692414834; }
    TB_HD static constexpr double w(int i) { return i == 1 ? 0.8888888888888888 : 0.5555555555555556; }
};

// trilinear hexahedron, vertices (-,-,-),(+,-,-),(+,+,-),(-,+,-),(-,-,+),(+,-,+),(+,+,+),(-,+,+)
template <int ORDER> struct Hex8 {
    static constexpr int NV = 8, NB = 8, NQ = ORDER * ORDER * ORDER;
    TB_HD static constexpr int sgn(int a, int d)
    {
        constexpr int S[3][8] = {{-1, 1, 1, -1, -1, 1, 1, -1}, {-1, -1, 1, 1, -1, -1, 1, 1}, {-1, -1, -1, -1, 1, 1, 1, 1}};
        return S[d][a];
    }
    // tensor rule, first coordinate fastest
    TB_HD static constexpr int qi(int q, int d) { return d == 0 ? q % ORDER : d == 1 ? (q / ORDER) % ORDER : q / (ORDER * ORDER); }
    TB_HD static constexpr double xi(int q, int d) { return Gauss<ORDER>::x(qi(q, d)); }
    TB_HD static constexpr double w(int q) { return Gauss<ORDER>::w(qi(q, 0)) * Gauss<ORDER>::w(qi(q, 1)) * Gauss<ORDER>::w(qi(q, 2)); }
    TB_HD static constexpr double fac(int q, int a, int d) { return 1.0 + sgn(a, d) * xi(q, d); }
    TB_HD static constexpr double N(int q, int a) { return 0.125 * fac(q, a, 0) * fac(q, a, 1) * fac(q, a, 2); }
    TB_HD static constexpr double dN(int q, int a, int d)
    {
        return 0.125 * (d == 0 ? sgn(a, 0) : fac(q, a, 0)) * (d == 1 ? sgn(a, 1) : fac(q, a, 1)) * (d == 2 ? sgn(a, 2) : fac(q, a, 2));
    }
    // geometry == field interpolation
    TB_HD static constexpr double M(int q, int a) { return N(q, a); }
    TB_HD static constexpr double dM(int q, int a, int d) { return dN(q, a, d); }
};

// ---- Hex27: triquadratic Lagrange field on the trilinear hexahedron, 3×3×3 Gauss points.  Ferrite Lagrange{RefHexahedron,2} local
// numbering: vertices, edges, faces, volume (the tensor index per direction below).  Used by the scalar Q2 forms (block-per-cell kernels).
struct Hex27 {
    static constexpr int NB = 27, NQ = 27, NV = 8;
    TB_HD static constexpr int tix(int a, int d)
    {
        constexpr int T[27][3] = {{0, 0, 0}, {2, 0, 0}, {2, 2, 0}, {0, 2, 0}, {0, 0, 2}, {2, 0, 2}, {2, 2, 2}, {0, 2, 2}, {1, 0, 0},
                                  {2, 1, 0}, {1, 2, 0}, {0, 1, 0}, {1, 0, 2}, {2, 1, 2}, {1, 2, 2}, {0, 1, 2}, {0, 0, 1}, {2, 0, 1},
                                  {2, 2, 1}, {0, 2, 1}, {1, 1, 0}, {1, 0, 1}, {2, 1, 1}, {1, 2, 1}, {0, 1, 1}, {1, 1, 2}, {1, 1, 1}};
        return T[a][d];
    }
    TB_HD static constexpr double q1(int i, double x) { return i == 0 ? 0.5 * x * (x - 1.0) : i == 1 ? (1.0 - x * x) : 0.5 * x * (x + 1.0); }
    TB_HD static constexpr double dq1(int i, double x) { return i == 0 ? x - 0.5 : i == 1 ? -2.0 * x : x + 0.5; }
    TB_HD static constexpr int qi(int q, int d) { return d == 0 ? q % 3 : d == 1 ? (q / 3) % 3 : q / 9; }
    TB_HD static constexpr double xi(int q, int d) { return Gauss<3>::x(qi(q, d)); }
    TB_HD static constexpr double w(int q) { return Gauss<3>::w(qi(q, 0)) * Gauss<3>::w(qi(q, 1)) * Gauss<3>::w(qi(q, 2)); }
    TB_HD static constexpr double N(int q, int a) { return q1(tix(a, 0), xi(q, 0)) * q1(tix(a, 1), xi(q, 1)) * q1(tix(a, 2), xi(q, 2)); }
    TB_HD static constexpr double dN(int q, int a, int d)
    {
        return (d == 0 ? dq1(tix(a, 0), xi(q, 0)) : q1(tix(a, 0), xi(q, 0))) * (d == 1 ? dq1(tix(a, 1), xi(q, 1)) : q1(tix(a, 1), xi(q, 1))) *
               (d == 2 ? dq1(tix(a, 2), xi(q, 2)) : q1(tix(a, 2), xi(q, 2)));
    }
    TB_HD static constexpr double M(int q, int a) { return Hex8<3>::N(q, a); }       // trilinear geometry at the same points
    TB_HD static constexpr double dM(int q, int a, int d) { return Hex8<3>::dN(q, a, d); }
};

// linear tetrahedron on the unit simplex; ORDER 1 → 1 point, ORDER 2 → 4-point degree-2 rule
template <int ORDER> struct Tet4 {
    static constexpr int NV = 4, NB = 4, NQ = ORDER == 1 ? 1 : 4;
    TB_HD static constexpr double xi(int q, int d)
    {
        return ORDER == 1 ? 0.25 : ((q == d + 1) ? 0.5854101966249685 : 0.1381966011250105);
    }
    TB_HD static constexpr double w(int) { return ORDER == 1 ? 1.0 / 6.0 : 1.0 / 24.0; }
    TB_HD static constexpr double N(int q, int a)
    {
        return a == 0 ? 1.0 - xi(q, 0) - xi(q, 1) - xi(q, 2) : xi(q, a - 1);
    }
    TB_HD static constexpr double dN(int, int a, int d) { return a == 0 ? -1.0 : (a - 1 == d ? 1.0 : 0.0); }
    TB_HD static constexpr double M(int q, int a) { return N(q, a); }
    TB_HD static constexpr double dM(int q, int a, int d) { return dN(q, a, d); }
};

// bilinear quadrilateral in the plane z = 0, vertices (-,-),(+,-),(+,+),(-,+) (Ferrite RefQuadrilateral).  The element is
// carried through the 3-D code with a trivial third reference direction: ∂x/∂ζ = e_z, ∂N/∂ζ = 0 — the Jacobian is
// block-diagonal, detJ is the 2-D determinant and the third row / column of the coefficient tensor never contributes.
template <int ORDER> struct Quad4 {
    static constexpr int NV = 4, NB = 4, NQ = ORDER * ORDER;
    TB_HD static constexpr int sgn(int a, int d)
    {
        constexpr int S[2][4] = {{-1, 1, 1, -1}, {-1, -1, 1, 1}};
        return S[d][a];
    }
    TB_HD static constexpr int qi(int q, int d) { return d == 0 ? q % ORDER : q / ORDER; }
    TB_HD static constexpr double xi(int q, int d) { return d < 2 ? Gauss<ORDER>::x(qi(q, d)) : 0.0; }
    TB_HD static constexpr double w(int q) { return Gauss<ORDER>::w(qi(q, 0)) * Gauss<ORDER>::w(qi(q, 1)); }
    TB_HD static constexpr double fac(int q, int a, int d) { return 1.0 + sgn(a, d) * xi(q, d); }
    TB_HD static constexpr double N(int q, int a) { return 0.25 * fac(q, a, 0) * fac(q, a, 1); }
    TB_HD static constexpr double dN(int q, int a, int d)
    {
        return d == 2 ? 0.0 : 0.25 * (d == 0 ? sgn(a, 0) : fac(q, a, 0)) * (d == 1 ? sgn(a, 1) : fac(q, a, 1));
    }
    TB_HD static constexpr double M(int q, int a) { return N(q, a); }
    TB_HD static constexpr double dM(int q, int a, int d) { return dN(q, a, d); }
};

struct Geom {
    double dOmega;  // detJ · w
    double Jinv[3][3];
};

// J, detJ, J⁻¹ at quadrature point Q of element type E.  x: vertex coordinates [NV][3].
// Returns detJ·w in g.dOmega; caller checks dOmega > 0.
template <class E, int Q>
__device__ __forceinline__ void geometry(const double (&x)[E::NV][3], Geom &g)
{
    double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
    for (int a = 0; a < E::NV; ++a)
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int k = 0; k < 3; ++k) J[i][k] += x[a][i] * E::dM(Q, a, k);
    const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
    const double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
    const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
    const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
    const double id = 1.0 / det;
    g.Jinv[0][0] = c00 * id; g.Jinv[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id; g.Jinv[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * id;
    g.Jinv[1][0] = c01 * id; g.Jinv[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id; g.Jinv[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id;
    g.Jinv[2][0] = c02 * id; g.Jinv[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * id; g.Jinv[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * id;
    g.dOmega = det * E::w(Q);
}

// dNₐ/dx = dNₐ/dξ · J⁻¹  (row vector times matrix), PR883.jl:287-290
template <class E, int Q>
__device__ __forceinline__ void mapped_gradients(const Geom &g, double (&grad)[E::NB][3])
{
#pragma unroll
    for (int a = 0; a < E::NB; ++a)
#pragma unroll
        for (int k = 0; k < 3; ++k)
            grad[a][k] = E::dN(Q, a, 0) * g.Jinv[0][k] + E::dN(Q, a, 1) * g.Jinv[1][k] + E::dN(Q, a, 2) * g.Jinv[2][k];
}

// Reference-element tables in constant memory.  The kernels loop over quadrature points at run time (keeps
// the register footprint to ONE point's worth of temporaries); the point index is wave-uniform, so every
// table entry is fetched with scalar loads and enters the FP64 FMAs as an SGPR operand.
template <class E> struct Tables {
    double N[E::NQ][E::NB];
    double dN[E::NQ][E::NB][3];
    double NN[E::NQ][E::NB * (E::NB + 1) / 2]; // NᵢNⱼ, upper triangle packed
    double w[E::NQ];
    double xi[E::NQ][6]; // ξ, η, ζ, ηζ, ζξ, ξη
};
template <class E> constexpr Tables<E> make_tables()
{
    Tables<E> t{};
    for (int q = 0; q < E::NQ; ++q) {
        t.w[q] = E::w(q);
        t.xi[q][0] = E::xi(q, 0); t.xi[q][1] = E::xi(q, 1); t.xi[q][2] = E::xi(q, 2);
        t.xi[q][3] = E::xi(q, 1) * E::xi(q, 2); t.xi[q][4] = E::xi(q, 2) * E::xi(q, 0); t.xi[q][5] = E::xi(q, 0) * E::xi(q, 1);
        int k = 0;
        for (int a = 0; a < E::NB; ++a) {
            t.N[q][a] = E::N(q, a);
            for (int d = 0; d < 3; ++d) t.dN[q][a][d] = E::dN(q, a, d);
            for (int b = a; b < E::NB; ++b) t.NN[q][k++] = E::N(q, a) * E::N(q, b);
        }
    }
    return t;
}
template <class E> __constant__ Tables<E> g_tables = make_tables<E>();

// Geometry of a cell in "modal" form.  Hexahedron: the trilinear map is
//   x(ξ) = c0 + c1 ξ + c2 η + c3 ζ + c4 ξη + c5 ηζ + c6 ζξ + c7 ξηζ      (cₖ = ⅛ Σₐ σₖ(a) Xₐ, a Walsh transform of the vertices)
// so J = Σₐ Xₐ ⊗ ∂Mₐ/∂ξ (PR883.jl:253-263) costs 27 FMAs per point instead of 72:
//   ∂x/∂ξ = c1 + c4 η + c6 ζ + c7 ηζ,  ∂x/∂η = c2 + c4 ξ + c5 ζ + c7 ζξ,  ∂x/∂ζ = c3 + c5 η + c6 ξ + c7 ξη.
// Tetrahedron: J is constant, stored directly (c[1..3] = edge vectors, c[0] = first vertex).
template <class E> struct GeoCoeffs { double c[E::NV][3]; };

template <int ORDER>
__device__ __forceinline__ void geo_prepare(const double (&x)[8][3], GeoCoeffs<Hex8<ORDER>> &g)
{
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        // stage ξ: (y,z) = (−,−): 0|1, (+,−): 3|2, (−,+): 4|5, (+,+): 7|6
        const double s0 = x[1][i] + x[0][i], d0 = x[1][i] - x[0][i];
        const double s1 = x[2][i] + x[3][i], d1 = x[2][i] - x[3][i];
        const double s2 = x[5][i] + x[4][i], d2 = x[5][i] - x[4][i];
        const double s3 = x[6][i] + x[7][i], d3 = x[6][i] - x[7][i];
        // stage η (per z)
        const double ss0 = s1 + s0, sd0 = s1 - s0, ds0 = d1 + d0, dd0 = d1 - d0; // z = −
        const double ss1 = s3 + s2, sd1 = s3 - s2, ds1 = d3 + d2, dd1 = d3 - d2; // z = +
        // stage ζ
        g.c[0][i] = 0.125 * (ss1 + ss0); g.c[3][i] = 0.125 * (ss1 - ss0);
        g.c[1][i] = 0.125 * (ds1 + ds0); g.c[6][i] = 0.125 * (ds1 - ds0);
        g.c[2][i] = 0.125 * (sd1 + sd0); g.c[5][i] = 0.125 * (sd1 - sd0);
        g.c[4][i] = 0.125 * (dd1 + dd0); g.c[7][i] = 0.125 * (dd1 - dd0);
    }
}
// Gauss coordinates of point q.  For the 2-point rule they are ±1/√3 selected by the bits of q — pure scalar
// arithmetic, so the geometry stage of a point does not wait for the scalar loads of the ∂N/∂ξ table.
template <int ORDER>
__device__ __forceinline__ void gauss_coords(const Tables<Hex8<ORDER>> &tb, int q, double (&c)[6])
{
    if constexpr (ORDER == 2) {
        constexpr double G = 0.5773502691896258, G2 = G * G;
        c[0] = (q & 1) ? G : -G; c[1] = (q & 2) ? G : -G; c[2] = (q & 4) ? G : -G;
        c[3] = (((q >> 1) ^ (q >> 2)) & 1) ? -G2 : G2;
        c[4] = (((q >> 2) ^ q) & 1) ? -G2 : G2;
        c[5] = ((q ^ (q >> 1)) & 1) ? -G2 : G2;
    } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) c[k] = tb.xi[q][k];
    }
}
template <int ORDER>
__device__ __forceinline__ void geo_jacobian(const GeoCoeffs<Hex8<ORDER>> &g, const Tables<Hex8<ORDER>> &tb, int q, double (&J)[3][3])
{
    double c[6];
    gauss_coords(tb, q, c);
    const double xi = c[0], eta = c[1], zeta = c[2], ez = c[3], zx = c[4], xe = c[5];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        J[i][0] = g.c[1][i] + g.c[4][i] * eta + g.c[6][i] * zeta + g.c[7][i] * ez;
        J[i][1] = g.c[2][i] + g.c[4][i] * xi + g.c[5][i] * zeta + g.c[7][i] * zx;
        J[i][2] = g.c[3][i] + g.c[5][i] * eta + g.c[6][i] * xi + g.c[7][i] * xe;
    }
}
template <int ORDER>
__device__ __forceinline__ void geo_position(const GeoCoeffs<Hex8<ORDER>> &g, const Tables<Hex8<ORDER>> &tb, int q, double (&xq)[3])
{
    double c[6];
    gauss_coords(tb, q, c);
    const double xi = c[0], eta = c[1], zeta = c[2], ez = c[3], zx = c[4], xe = c[5];
    const double xez = xi * ez;
#pragma unroll
    for (int i = 0; i < 3; ++i)
        xq[i] = g.c[0][i] + g.c[1][i] * xi + g.c[2][i] * eta + g.c[3][i] * zeta + g.c[4][i] * xe + g.c[5][i] * ez + g.c[6][i] * zx + g.c[7][i] * xez;
}

template <int ORDER>
__device__ __forceinline__ void geo_prepare(const double (&x)[4][3], GeoCoeffs<Tet4<ORDER>> &g)
{
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        g.c[0][i] = x[0][i];
        g.c[1][i] = x[1][i] - x[0][i]; g.c[2][i] = x[2][i] - x[0][i]; g.c[3][i] = x[3][i] - x[0][i];
    }
}
template <int ORDER>
__device__ __forceinline__ void geo_jacobian(const GeoCoeffs<Tet4<ORDER>> &g, const Tables<Tet4<ORDER>> &, int, double (&J)[3][3])
{
#pragma unroll
    for (int i = 0; i < 3; ++i) { J[i][0] = g.c[1][i]; J[i][1] = g.c[2][i]; J[i][2] = g.c[3][i]; }
}
template <int ORDER>
__device__ __forceinline__ void geo_position(const GeoCoeffs<Tet4<ORDER>> &g, const Tables<Tet4<ORDER>> &tb, int q, double (&xq)[3])
{
#pragma unroll
    for (int i = 0; i < 3; ++i) xq[i] = g.c[0][i] + g.c[1][i] * tb.xi[q][0] + g.c[2][i] * tb.xi[q][1] + g.c[3][i] * tb.xi[q][2];
}

// Quadrilateral: x(ξ,η) = c0 + c1 ξ + c2 η + c3 ξη
template <int ORDER>
__device__ __forceinline__ void geo_prepare(const double (&x)[4][3], GeoCoeffs<Quad4<ORDER>> &g)
{
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double s0 = x[1][i] + x[0][i], d0 = x[1][i] - x[0][i]; // y = −
        const double s1 = x[2][i] + x[3][i], d1 = x[2][i] - x[3][i]; // y = +
        g.c[0][i] = 0.25 * (s1 + s0); g.c[2][i] = 0.25 * (s1 - s0);
        g.c[1][i] = 0.25 * (d1 + d0); g.c[3][i] = 0.25 * (d1 - d0);
    }
}
template <int ORDER>
__device__ __forceinline__ void geo_jacobian(const GeoCoeffs<Quad4<ORDER>> &g, const Tables<Quad4<ORDER>> &tb, int q, double (&J)[3][3])
{
    const double xi = tb.xi[q][0], eta = tb.xi[q][1];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        J[i][0] = g.c[1][i] + g.c[3][i] * eta;
        J[i][1] = g.c[2][i] + g.c[3][i] * xi;
        J[i][2] = i == 2 ? 1.0 : 0.0;
    }
}
template <int ORDER>
__device__ __forceinline__ void geo_position(const GeoCoeffs<Quad4<ORDER>> &g, const Tables<Quad4<ORDER>> &tb, int q, double (&xq)[3])
{
    const double xi = tb.xi[q][0], eta = tb.xi[q][1];
#pragma unroll
    for (int i = 0; i < 3; ++i) xq[i] = g.c[0][i] + g.c[1][i] * xi + g.c[2][i] * eta + g.c[3][i] * (xi * eta);
}

// J, detJ·w, J⁻¹ at run-time quadrature point q (geometry interpolation == field interpolation, first-order cells)
template <class E>
__device__ __forceinline__ void geometry_rt(const Tables<E> &tb, int q, const GeoCoeffs<E> &gc, Geom &g)
{
    double J[3][3];
    geo_jacobian(gc, tb, q, J);
    const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
    const double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
    const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
    const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
    const double id = 1.0 / det;
    g.Jinv[0][0] = c00 * id; g.Jinv[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id; g.Jinv[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * id;
    g.Jinv[1][0] = c01 * id; g.Jinv[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id; g.Jinv[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id;
    g.Jinv[2][0] = c02 * id; g.Jinv[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * id; g.Jinv[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * id;
    if constexpr (std::is_same<E, Hex8<2>>::value) g.dOmega = det; // Gauss weights of the 2-point rule are all 1
    else g.dOmega = det * tb.w[q];
}

// compile-time loop over quadrature points
template <int Q, int NQ, class F>
__device__ __forceinline__ void for_each_qp(F &&f)
{
    if constexpr (Q < NQ) {
        f(std::integral_constant<int, Q>{});
        for_each_qp<Q + 1, NQ>(f);
    }
}

} // namespace tbk
