// tb_mech_common.hpp — reference-element tables and kernel-argument blocks shared by the mechanics kernels (tb_mechanics.hip, tb_mech_split.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "tb_internal.h"
#include "tb_material.hpp"

namespace tb {
using namespace tbk;

// ---- reference-element tables for the two supported vector fields (geometry: trilinear hexahedron) ----
struct G3 {
    __host__ __device__ static constexpr double x(int i) { return i == 0 ? -0.7745966692414834 : i == 1 ? 0.0 : 0.7745966692414834; }
    __host__ __device__ static constexpr double w(int i) { return i == 1 ? 0.8888888888888888 : 0.5555555555555556; }
};
struct G2 {
    __host__ __device__ static constexpr double x(int i) { return i == 0 ? -0.5773502691896258 : 0.5773502691896258; }
    __host__ __device__ static constexpr double w(int) { return 1.0; }
};

__host__ __device__ constexpr int hex_sgn(int a, int d)
{
    constexpr int S[3][8] = {{-1, 1, 1, -1, -1, 1, 1, -1}, {-1, -1, 1, 1, -1, -1, 1, 1}, {-1, -1, -1, -1, 1, 1, 1, 1}};
    return S[d][a];
}
// Ferrite Lagrange{RefHexahedron,2} local numbering: vertices, edges, faces, volume → tensor index per direction
__host__ __device__ constexpr int hex27_tix(int a, int d)
{
    constexpr int T[27][3] = {{0, 0, 0}, {2, 0, 0}, {2, 2, 0}, {0, 2, 0}, {0, 0, 2}, {2, 0, 2}, {2, 2, 2}, {0, 2, 2}, {1, 0, 0},
                              {2, 1, 0}, {1, 2, 0}, {0, 1, 0}, {1, 0, 2}, {2, 1, 2}, {1, 2, 2}, {0, 1, 2}, {0, 0, 1}, {2, 0, 1},
                              {2, 2, 1}, {0, 2, 1}, {1, 1, 0}, {1, 0, 1}, {2, 1, 1}, {1, 2, 1}, {0, 1, 1}, {1, 1, 2}, {1, 1, 1}};
    return T[a][d];
}
// inverse of hex27_tix: tensor index t₀ + 3 t₁ + 9 t₂ → Ferrite node
struct Hex27Nodes { int v[27]; };
constexpr Hex27Nodes make_hex27_nodes()
{
    Hex27Nodes n{};
    for (int a = 0; a < 27; ++a) n.v[hex27_tix(a, 0) + 3 * hex27_tix(a, 1) + 9 * hex27_tix(a, 2)] = a;
    return n;
}
static __constant__ int g_hex27_node[27] = {make_hex27_nodes().v[0], make_hex27_nodes().v[1], make_hex27_nodes().v[2], make_hex27_nodes().v[3], make_hex27_nodes().v[4],
                                     make_hex27_nodes().v[5], make_hex27_nodes().v[6], make_hex27_nodes().v[7], make_hex27_nodes().v[8], make_hex27_nodes().v[9],
                                     make_hex27_nodes().v[10], make_hex27_nodes().v[11], make_hex27_nodes().v[12], make_hex27_nodes().v[13], make_hex27_nodes().v[14],
                                     make_hex27_nodes().v[15], make_hex27_nodes().v[16], make_hex27_nodes().v[17], make_hex27_nodes().v[18], make_hex27_nodes().v[19],
                                     make_hex27_nodes().v[20], make_hex27_nodes().v[21], make_hex27_nodes().v[22], make_hex27_nodes().v[23], make_hex27_nodes().v[24],
                                     make_hex27_nodes().v[25], make_hex27_nodes().v[26]};
// Tensor-order layout of the stored element matrices of the triquadratic field (element strategy, sum-factorised kernels; round 5):
//   Kₑ[(a,c)][(b,d)] sits at row 3·tix(a) + c, column cb(b) + 9·d with tix(a) = a₀ + 3a₁ + 9a₂ and cb(b) = 27·b₂ + 3·b₀ + b₁ (tensor indices of
// the Ferrite nodes) — the order in which the lanes of the contraction produce their entries, so that 27 consecutive lanes store 27 consecutive
// doubles (in Ferrite order a wave-instruction of 64 stores touched 64 different 64-byte segments, and the address path, not the arithmetic,
// bounded the kernel).  The gather undoes it through these two 27-entry tables, packed six bits per entry into three 64-bit constants.
constexpr uint64_t pack27(int w, bool colbase)
{
    uint64_t v = 0;
    for (int i = 0; i < 10; ++i) {
        const int a = 10 * w + i;
        if (a < 27) {
            const int t0 = hex27_tix(a, 0), t1 = hex27_tix(a, 1), t2 = hex27_tix(a, 2);
            v |= (uint64_t)(colbase ? 27 * t2 + 3 * t0 + t1 : t0 + 3 * t1 + 9 * t2) << (6 * i);
        }
    }
    return v;
}
__host__ __device__ __forceinline__ int unpack27(int a, bool colbase)
{
    const uint64_t w = a < 10 ? pack27(0, colbase) : a < 20 ? pack27(1, colbase) : pack27(2, colbase);
    return (int)((w >> (6 * (a % 10))) & 63);
}
__host__ __device__ __forceinline__ int tix27(int a) { return unpack27(a, false); } // Ferrite node → a₀ + 3a₁ + 9a₂
__host__ __device__ __forceinline__ int cb27(int b) { return unpack27(b, true); }   // Ferrite node → 27·b₂ + 3·b₀ + b₁
// symmetric-packed element matrix of the 27-node vector field: 378 node blocks (i ≤ j by rank) of 9 doubles
__host__ __device__ constexpr int symblk(int i, int j) { return i * 27 - i * (i - 1) / 2 + (j - i); }
constexpr int KE_SYM = 378 * 9;
__host__ __device__ constexpr double quad1d(int i, double x) { return i == 0 ? 0.5 * x * (x - 1.0) : i == 1 ? (1.0 - x * x) : 0.5 * x * (x + 1.0); }
__host__ __device__ constexpr double dquad1d(int i, double x) { return i == 0 ? x - 0.5 : i == 1 ? -2.0 * x : x + 0.5; }

template <int NB_, int NQ1D_, int PB_, int THREADS_, int WAVES_> struct VecField {
    static constexpr int NB = NB_, NQ1D = NQ1D_, NQ = NQ1D_ * NQ1D_ * NQ1D_, PB = PB_, THREADS = THREADS_, ND = 3 * NB_;
    static constexpr int WAVES = WAVES_; // min waves per SIMD the register allocator must leave room for (measured best)
    __host__ __device__ static constexpr double gx(int i) { return NQ1D == 2 ? G2::x(i) : G3::x(i); }
    __host__ __device__ static constexpr double gw(int i) { return NQ1D == 2 ? G2::w(i) : G3::w(i); }
    __host__ __device__ static constexpr double xi(int q, int d) { return gx(d == 0 ? q % NQ1D : d == 1 ? (q / NQ1D) % NQ1D : q / (NQ1D * NQ1D)); }
    __host__ __device__ static constexpr double w(int q) { return gw(q % NQ1D) * gw((q / NQ1D) % NQ1D) * gw(q / (NQ1D * NQ1D)); }
    __host__ __device__ static constexpr double dM(int q, int a, int d) // trilinear geometry
    {
        return 0.125 * (d == 0 ? hex_sgn(a, 0) : 1.0 + hex_sgn(a, 0) * xi(q, 0)) * (d == 1 ? hex_sgn(a, 1) : 1.0 + hex_sgn(a, 1) * xi(q, 1)) *
               (d == 2 ? hex_sgn(a, 2) : 1.0 + hex_sgn(a, 2) * xi(q, 2));
    }
    __host__ __device__ static constexpr double dN(int q, int a, int d)
    {
        if (NB == 8) return dM(q, a, d);
        return (d == 0 ? dquad1d(hex27_tix(a, 0), xi(q, 0)) : quad1d(hex27_tix(a, 0), xi(q, 0))) *
               (d == 1 ? dquad1d(hex27_tix(a, 1), xi(q, 1)) : quad1d(hex27_tix(a, 1), xi(q, 1))) *
               (d == 2 ? dquad1d(hex27_tix(a, 2), xi(q, 2)) : quad1d(hex27_tix(a, 2), xi(q, 2)));
    }
};
using Q1Vec = VecField<8, 2, 1, 64, 2>;    // 64 threads: one per node pair
using Q2Vec = VecField<27, 3, 3, 256, 3>;  // 243 threads: node a × 9 groups of 3 nodes b

template <class FE> struct MechTables {
    double dN[FE::NQ][FE::NB][3];
    double dM[FE::NQ][8][3];
    double M[FE::NQ][8]; // first-order shape values (nodal microstructure fields)
    double w[FE::NQ];
};
template <class FE> constexpr MechTables<FE> make_mech_tables()
{
    MechTables<FE> t{};
    for (int q = 0; q < FE::NQ; ++q) {
        t.w[q] = FE::w(q);
        for (int a = 0; a < FE::NB; ++a)
            for (int d = 0; d < 3; ++d) t.dN[q][a][d] = FE::dN(q, a, d);
        for (int a = 0; a < 8; ++a) {
            t.M[q][a] = 0.125 * (1.0 + hex_sgn(a, 0) * FE::xi(q, 0)) * (1.0 + hex_sgn(a, 1) * FE::xi(q, 1)) * (1.0 + hex_sgn(a, 2) * FE::xi(q, 2));
            for (int d = 0; d < 3; ++d) t.dM[q][a][d] = FE::dM(q, a, d);
        }
    }
    return t;
}
template <class FE> __constant__ MechTables<FE> g_mech_tables = make_mech_tables<FE>();

struct MechMesh {
    const double *xyz;
    const int32_t *conn;
    const int32_t *cell_dofs;
    const double *fsn_field; // per cell and geometric node: f,s,n (9 doubles); NULL → constant frame of the material
    const double *act_field; // per cell and geometric node: calcium-driven state multiplying the active tension; NULL → 1
    const double *qp_act;    // condensed internal variable: per quadrature point (a, b) — P += a ∂λ/∂F, 𝔸 += a ∂²λ/∂F² + b ∂λ/∂F ⊗ ∂λ/∂F; NULL → none
    int qp_stride;           // 2, or 5 in the rate-coupled form: (a, b, c·w) with the non-symmetric term 𝔸 += ∂λ/∂F ⊗ (c·w ⊗ f₀), w ⊗ f₀ = ∂²λ/∂F² : Ḟ
    const double *cell_xyz;  // vertex coordinates per cell (24 doubles, cell-major): the sum-factorised tangent kernel reads them with wave-uniform addresses
    int64_t cell0;           // first cell of the launch (chunked linearisation); 0 otherwise
#ifdef TB_ABLATION
    long long *prof;         // TB_PROF_STAMPS: 16 phase time stamps of every 256th workgroup
#endif
};


} // namespace tb
