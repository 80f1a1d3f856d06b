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
