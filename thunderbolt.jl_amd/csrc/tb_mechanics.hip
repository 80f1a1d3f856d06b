// tb_mechanics.hip — quasi-static hyperelastic residual / tangent element kernels (Holzapfel–Ogden 2009),
// vector-valued Q1 and Q2 Lagrange fields on trilinear hexahedra.
//
// Restates src/modeling/solid/elements.jl:177-225 (K and r), :227-273 (K), :275-313 (r):
//   per quadrature point  ∇u = Σ uₑᵢ ∇δuᵢ,  F = I + ∇u,  (P, 𝔸) = material_routine(F)   (materials.jl:442-453,1025-1040)
//   rₑ[i] += ∇δuᵢ ⊡ P dΩ,   Kₑ[i,j] += (∇δuᵢ ⊡ 𝔸) ⊡ ∇δuⱼ dΩ,   dof i = 3a + c ↔ ∇δuᵢ = e_c ⊗ ∇Nₐ.
// The cell loop / load_element_unknowns! / assemble! belong to FerriteOperators (third party); call sites
// src/solver/nonlinear/newton_raphson.jl:234-238, src/solver/time/homotopy.jl:61-67.
//
// One workgroup per cell.  The tangent is contracted in two steps, T[a][c][d][l] = Σ_k ∇Nₐ[k] 𝔸[c][k][d][l] and then
// Kₑ[(a,c)][(b,d)] += Σ_l T[a][c][d][l] ∇N_b[l] — 27·NB² FMAs per point instead of 81·NB².
//  * Q2 (default): the second step is a GEMM per component d, C_d[(a,c)][b] = Σ_(q,l) T_d[(a,c)][(q,l)] · ∇N[(q,l)][b]
//    (M = 81, N = 27, K = 81), run on v_mfma_f64_16x16x4_f64: 6×2 tiles × 3 d, 21 k-steps.  The A operand (one T entry
//    per lane) is formed in registers from 𝔸 and ∇N straight out of LDS, so T is never staged.  FP64 MFMA is not faster
//    than FP64 vector FMAs on CDNA4 (measured 48 vs 70 TFLOP/s, scripts/microbench/mfma_f64.hip) — the point is operand
//    reuse: the vector version needs one LDS double per 2.25 FMAs and is LDS-bandwidth-bound at ≈20 % of peak.
//  * Q1 and the comparison build (TB_MECH_MFMA=0): thread (a, b-group) keeps 3×3 blocks in registers, T staged in LDS.
#include <hip/hip_runtime.h>

#include <cstring>

#include <algorithm>

#include "tb_internal.h"
#include "tb_energy.hpp"
#include "tb_material.hpp"
#include "tb_math.hpp"
#include "tb_mech_common.hpp"
#include "tb_mech_split.hpp"

#ifdef TB_ABLATION
#define TB_IF_ABLATION(...) __VA_ARGS__
#else
#define TB_IF_ABLATION(...)
#endif

namespace tb {
using namespace tbk;

// position of column dof(b,0) inside row dof(a,0), per cell and node pair (the three component rows of a node
// share one column set, so the same position serves rows +1, +2 and columns +1, +2)
__global__ void k_build_blockpos(const int32_t *__restrict__ cell_dofs, int64_t n_cells, int nb, const int64_t *__restrict__ rowptr,
                                 const int32_t *__restrict__ colidx, uint16_t *__restrict__ pos, Status *st)
{
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= n_cells * nb * nb) return;
    const int64_t cell = tid / (nb * nb);
    const int a = (int)(tid % (nb * nb)) / nb, b = (int)(tid % nb);
    const int32_t *d = cell_dofs + cell * 3 * nb;
    const int32_t row = d[3 * a], col = d[3 * b];
    const int64_t lo0 = rowptr[row], hi0 = rowptr[row + 1];
    int64_t lo = lo0, hi = hi0;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (colidx[mid] < col) lo = mid + 1; else hi = mid;
    }
    if (lo >= hi0 || colidx[lo] != col || lo - lo0 > 0xFFFF) { st->pattern_missing = 1; st->cell = cell; lo = lo0; }
    pos[tid] = (uint16_t)(lo - lo0);
}

typedef double mfma_d4 __attribute__((ext_vector_type(4)));
#ifndef TB_KE_DMAJOR
#define TB_KE_DMAJOR 0
#endif
constexpr bool KE_DMAJOR = TB_KE_DMAJOR; // stored-Kₑ row layout of the matrix-core kernel: [d][b] or [b][d]

// AD: the material is any energy of tb_energy.hpp, differentiated per pair of components of F by hyper-dual evaluation (the
// reference's Tensors.hessian path); !AD: Holzapfel–Ogden 2009 + SimpleCompressionPenalty with the hand-derived routines.
// CT: contraction of the tangent — 0 vector FMAs over the quadrature points (Q1; comparison build), 1 matrix cores (v_mfma_f64_16x16x4,
// symmetric tiles), 2 sum-factorised over the tensor-product structure of the triquadratic basis and the 3×3×3 Gauss rule (Q2 default)
template <class FE, bool NEED_K, bool NEED_R, int CT, bool AD>
__global__ void __launch_bounds__(FE::THREADS, FE::WAVES)
k_hyperelastic(MechMesh m, HOParams mat, EnergyParams en, const int32_t *__restrict__ list, const double *__restrict__ u, double *__restrict__ nz,
               double *__restrict__ r, const int64_t *__restrict__ rowptr, const uint16_t *__restrict__ blockpos, int atomic /*0 rmw, 1 atomic, 2 store Kₑ/rₑ*/,
               double *__restrict__ ke, double *__restrict__ re, Status *st, const uint8_t *__restrict__ rank27 /*non-NULL: symmetric-packed Kₑ (matrix-core path)*/)
{
    constexpr int NB = FE::NB, NQ = FE::NQ, ND = FE::ND, PB = FE::PB, T = FE::THREADS, NG = NB / PB;
    constexpr bool MFMA = CT == 1, SF = CT == 2;
    // sum-factorised tangent: the mapped gradients ∇N = ∂̂N·J⁻¹ are never formed.  ∇u = Ĥ·J⁻¹ with Ĥ[c][s] = Σₐ uₐ[c] ∂̂ₛNₐ (reference gradients from the
    // 1-D factors in registers, concurrent with the Jacobians), the tangent and the stress are pulled back to the reference cell (stage 0), and the
    // residual contracts the pulled-back stress with the reference gradients: two barriers and the 729-task gradient pass less per cell
    constexpr bool REFGRAD = (NEED_K && SF) || (!NEED_K && NEED_R && NB == 27 && NQ == 27 && T == 256); // (the residual-only kernel of the quadratic field too)
    const MechTables<FE> &tb = g_mech_tables<FE>;
    const int64_t cell = list ? list[blockIdx.x] : m.cell0 + blockIdx.x;
    const int tid = threadIdx.x;
#ifdef TB_ABLATION
#define TB_MS(k) do { if (m.prof && tid == 0 && (blockIdx.x & 255) == 7) m.prof[(blockIdx.x >> 8) * 16 + (k)] = wall_clock64(); } while (0)
#else
#define TB_MS(k) do { } while (0)
#endif
    TB_MS(0);
    __shared__ double s_ue[ND], s_x[24], s_JI[NQ][10], s_P[NQ][9];
    // 𝔸·dΩ per point and the mapped gradients share one block: once the contraction is done it stages the symmetric-packed Kₑ for a coalesced store
    // (dynamic LDS: with the stage buffers of the sum-factorised contraction the kernel's block exceeds the 64 KB a static allocation may have)
    constexpr int AN = (NEED_K ? NQ : 1) * 81;
    extern __shared__ double s_AG[];
    double (*s_A)[81] = reinterpret_cast<double (*)[81]>(s_AG);
    double (*s_G)[NB][3] = reinterpret_cast<double (*)[NB][3]>(s_AG + AN);
    __shared__ uint8_t s_rank[32];
    // phase A scratch (common blocks + F) and phase B's double-buffered T share one region
    constexpr int TC_SIZE = (NEED_K && CT == 0 && 2 * NB * 27 > NQ * HOC_SIZE) ? 2 * NB * 27 : NQ * HOC_SIZE;
    __shared__ double s_Ji[REFGRAD ? NQ : 1][9]; // J⁻¹ kept for the pull-backs of 𝔸 and P (the slots of s_JI carry F)
    __shared__ double s_TC[TC_SIZE];
    double (*s_T)[NB][27] = reinterpret_cast<double (*)[NB][27]>(s_TC);
    double (*s_C)[HOC_SIZE] = reinterpret_cast<double (*)[HOC_SIZE]>(s_TC);
    __shared__ int32_t s_dof[ND];

    __shared__ uint8_t s_node[32], s_tix[32]; // tensor index t₀ + 3t₁ + 9t₂ → Ferrite node and back (LDS copies: a load from the constant table in the
                                              // middle of the kernel queues behind the element-matrix stores of the other workgroups, ≈ 1.5 µs)
    if constexpr (REFGRAD) {
        // load_element_unknowns! (elements.jl:125-132): dofs → u is two dependent trips; the Jacobians of the points are computed inside the second
        // one — vertex coordinates from the cell-major array with wave-uniform addresses (scalar loads: one trip, no connectivity chase, no LDS
        // staging), ∂M/∂ξ and the weights from the vertex signs and the 1-D rule in registers (no table loads with lane-varying addresses)
        static_assert(ND <= T && NB == 27, "one lane per element unknown");
        int32_t dof = 0;
        double uv = 0.0;
        if (tid < ND) { dof = m.cell_dofs[cell * ND + tid]; uv = u[dof]; }
        if (tid < 27) { const int a = g_hex27_node[tid]; s_node[tid] = (uint8_t)a; s_tix[a] = (uint8_t)tid; }
        if (tid < NQ) { // A1: J, J⁻¹, dΩ per point (PR883.jl:253-263,367-387)
            const int q = tid, qd[3] = {q % 3, (q / 3) % 3, q / 9};
            double mm_[3], pp_[3], wq = 1.0;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const double xi = qd[d] == 0 ? G3::x(0) : qd[d] == 1 ? G3::x(1) : G3::x(2);
                mm_[d] = 1.0 - xi; pp_[d] = 1.0 + xi;
                wq *= qd[d] == 1 ? G3::w(1) : G3::w(0);
            }
            const double *X = m.cell_xyz + cell * 24;
            double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
            for (int a = 0; a < 8; ++a) {
                const double f0 = hex_sgn(a, 0) > 0 ? pp_[0] : mm_[0], f1 = hex_sgn(a, 1) > 0 ? pp_[1] : mm_[1], f2 = hex_sgn(a, 2) > 0 ? pp_[2] : mm_[2];
                const double dm[3] = {0.125 * hex_sgn(a, 0) * f1 * f2, 0.125 * hex_sgn(a, 1) * f0 * f2, 0.125 * hex_sgn(a, 2) * f0 * f1};
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const double xa = X[3 * a + i];
#pragma unroll
                    for (int k = 0; k < 3; ++k) J[i][k] += xa * dm[k];
                }
            }
            const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1], c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2], c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
            const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02, id = 1.0 / det;
            double o[10];
            o[0] = c00 * id; o[1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id; o[2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * id;
            o[3] = c01 * id; o[4] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id; o[5] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id;
            o[6] = c02 * id; o[7] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * id; o[8] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * id;
            o[9] = det * wq;
            if (!(o[9] > 0.0)) { st->neg_detj = 1; st->cell = cell; }
#pragma unroll
            for (int e = 0; e < 9; ++e) s_Ji[q][e] = o[e];
            s_JI[q][9] = o[9];
        }
        if (tid < ND) { s_dof[tid] = dof; s_ue[tid] = uv; }
        __syncthreads();
        TB_MS(1);
    } else {
    // load_element_unknowns! (elements.jl:125-132) + cell coordinates
    for (int i = tid; i < ND; i += T) {
        const int32_t d = m.cell_dofs[cell * ND + i];
        s_dof[i] = d;
        s_ue[i] = u[d];
    }
    for (int i = tid; i < 24; i += T) s_x[i] = m.xyz[3 * (int64_t)m.conn[cell * 8 + i / 3] + i % 3];
    if (NEED_K && MFMA && rank27 && tid < 32) s_rank[tid] = rank27[cell * 32 + tid];
    __syncthreads();
    TB_MS(1);
    }

    // A1: J, J⁻¹, dΩ per point (PR883.jl:253-263,367-387)
    if (!REFGRAD && tid < NQ) {
        const int q = tid;
        double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int k = 0; k < 3; ++k) J[i][k] += s_x[3 * a + i] * tb.dM[q][a][k];
        const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1], c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2], c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
        const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02, id = 1.0 / det;
        double *o = s_JI[q];
        o[0] = c00 * id; o[1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id; o[2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * id;
        o[3] = c01 * id; o[4] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id; o[5] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id;
        o[6] = c02 * id; o[7] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * id; o[8] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * id;
        o[9] = det * tb.w[q];
        if (!(o[9] > 0.0)) { st->neg_detj = 1; st->cell = cell; }
        if constexpr (NEED_K && SF) {
#pragma unroll
            for (int e = 0; e < 9; ++e) s_Ji[q][e] = o[e];
        }
    }
    if constexpr (REFGRAD) {
        // A2': Ĥ[c][s] = Σₐ uₐ[c] ∂̂ₛNₐ(ξ_q), one lane per (point, c, s) — needs the element unknowns only, so it shares the phase with A1
        if (tid < NQ * 9) {
            const int q = tid / 9, cs = tid - 9 * q, c = cs / 3, s_ = cs - 3 * c;
            const int q0 = q % 3, q1 = (q / 3) % 3, q2 = q / 9;
            auto PHq = [](int i, int qq) { return qq == 0 ? quad1d(i, G3::x(0)) : qq == 1 ? quad1d(i, G3::x(1)) : quad1d(i, G3::x(2)); };
            auto DPq = [](int i, int qq) { return qq == 0 ? dquad1d(i, G3::x(0)) : qq == 1 ? dquad1d(i, G3::x(1)) : dquad1d(i, G3::x(2)); };
            double f0[3], f1[3], f2[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                f0[i] = s_ == 0 ? DPq(i, q0) : PHq(i, q0);
                f1[i] = s_ == 1 ? DPq(i, q1) : PHq(i, q1);
                f2[i] = s_ == 2 ? DPq(i, q2) : PHq(i, q2);
            }
            double h = 0.0;
#pragma unroll
            for (int a2 = 0; a2 < 3; ++a2) {
                double h1 = 0.0;
#pragma unroll
                for (int a1 = 0; a1 < 3; ++a1) {
                    double h0 = 0.0;
#pragma unroll
                    for (int a0 = 0; a0 < 3; ++a0) h0 += f0[a0] * s_ue[3 * make_hex27_nodes().v[(a0 + 3 * a1 + 9 * a2) % (NB == 27 ? 27 : 1)] + c];
                    h1 += f1[a1] * h0;
                }
                h += f2[a2] * h1;
            }
            s_P[q][cs] = h; // the stress block is free until A3c
        }
    } else {
    __syncthreads();
    // A2: mapped gradients ∇Nₐ = ∂Nₐ/∂ξ · J⁻¹ for every (point, node)
    for (int idx = tid; idx < NQ * NB; idx += T) {
        const int q = idx / NB, a = idx % NB;
        const double *ji = s_JI[q];
        const double d0 = tb.dN[q][a][0], d1 = tb.dN[q][a][1], d2 = tb.dN[q][a][2];
#pragma unroll
        for (int k = 0; k < 3; ++k) s_G[q][a][k] = d0 * ji[k] + d1 * ji[3 + k] + d2 * ji[6 + k];
    }
    __syncthreads();
    // A3a: F = I + ∇u, one lane per (point, component c, direction k)
    for (int t = tid; t < NQ * 9; t += T) {
        const int q = t / 9, ck = t % 9, c = ck / 3, k = ck % 3;
        double v = c == k ? 1.0 : 0.0;
        for (int a = 0; a < NB; ++a) v += s_ue[3 * a + c] * s_G[q][a][k];
        s_JI[q][ck] = v; // J⁻¹ is dead after A2: its slots carry F from here on
    }
    }
    __syncthreads();
    TB_MS(2);
    // F = I + Ĥ·J⁻¹ by the lane of the point (the lanes that evaluate the material next), into the slots of s_JI like A3a
    auto deformation_gradient = [&](int q) {
        const double *ji = s_Ji[q], *H = s_P[q];
        double Fv[9];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int k = 0; k < 3; ++k) Fv[3 * c + k] = (c == k ? 1.0 : 0.0) + H[3 * c] * ji[k] + H[3 * c + 1] * ji[3 + k] + H[3 * c + 2] * ji[6 + k];
#pragma unroll
        for (int e = 0; e < 9; ++e) s_JI[q][e] = Fv[e];
    };
    if constexpr (AD) {
        // A3b': frame and active tension of every point (constant, or interpolated nodal data like OrthotropicMicrostructureModel)
        if (tid < NQ) {
            if constexpr (REFGRAD) deformation_gradient(tid);
            double f[3] = {mat.f[0], mat.f[1], mat.f[2]}, sv[3] = {mat.s[0], mat.s[1], mat.s[2]}, n[3] = {mat.n[0], mat.n[1], mat.n[2]};
            if (m.fsn_field) {
                for (int d = 0; d < 3; ++d) { f[d] = 0.0; sv[d] = 0.0; n[d] = 0.0; }
                const double *fc = m.fsn_field + cell * 72;
                for (int a = 0; a < 8; ++a) {
                    const double Na = tb.M[tid][a];
                    for (int d = 0; d < 3; ++d) { f[d] += Na * fc[9 * a + d]; sv[d] += Na * fc[9 * a + 3 + d]; n[d] += Na * fc[9 * a + 6 + d]; }
                }
                ho_orthonormal_frame(f, sv, n);
            }
            double ta = en.Ta;
            if (m.act_field) { double ca = 0.0; for (int a = 0; a < 8; ++a) ca += tb.M[tid][a] * m.act_field[cell * 8 + a]; ta *= ca; }
            double *o = s_C[tid];
            for (int d = 0; d < 3; ++d) { o[d] = f[d]; o[3 + d] = sv[d]; o[6 + d] = n[d]; }
            o[9] = ta;
            o[10] = 0.0;
            if (m.qp_act) { o[9] = m.qp_act[m.qp_stride * (cell * NQ + tid)]; o[10] = m.qp_act[m.qp_stride * (cell * NQ + tid) + 1]; }
        }
        __syncthreads();
        // A3c': one lane per (point, pair of components of F): Ψ.a = P_m, Ψ.ab = 𝔸_mn = 𝔸_nm
        constexpr int NP = NEED_K ? 45 : 9;
        for (int t = tid; t < NQ * NP; t += T) {
            const int q = t / NP, pr = t - q * NP;
            int mm = pr, nn = pr;
            if constexpr (NEED_K) pair_components(pr, mm, nn);
            const double *o = s_C[q];
            const double f[3] = {o[0], o[1], o[2]}, sv[3] = {o[3], o[4], o[5]}, n[3] = {o[6], o[7], o[8]};
            const HD r = energy_pair(en, s_JI[q], mm, nn, f, sv, n, o[9]);
            const double dO = s_JI[q][9];
            if constexpr (NEED_K) {
                double ab = r.ab;
                if (o[10] != 0.0) { // condensed internal variable: + b (∂λ/∂F)_m (∂λ/∂F)_n, ∂λ/∂F = (F f₀) ⊗ f₀ / λ
                    const double *Fq = s_JI[q];
                    const double g0 = Fq[0] * f[0] + Fq[1] * f[1] + Fq[2] * f[2], g1 = Fq[3] * f[0] + Fq[4] * f[1] + Fq[5] * f[2],
                                 g2 = Fq[6] * f[0] + Fq[7] * f[1] + Fq[8] * f[2];
                    const double gv[3] = {g0, g1, g2};
                    ab += o[10] * (gv[mm / 3] * f[mm % 3]) * (gv[nn / 3] * f[nn % 3]) / (g0 * g0 + g1 * g1 + g2 * g2);
                }
                if (m.qp_act && m.qp_stride == 5) { // rate-coupled internal variable: the non-symmetric rank-one term, both orderings of the pair
                    const double *Fq = s_JI[q], *cw = m.qp_act + 5 * (cell * NQ + q) + 2;
                    const double g0 = Fq[0] * f[0] + Fq[1] * f[1] + Fq[2] * f[2], g1 = Fq[3] * f[0] + Fq[4] * f[1] + Fq[5] * f[2],
                                 g2 = Fq[6] * f[0] + Fq[7] * f[1] + Fq[8] * f[2];
                    const double gv[3] = {g0, g1, g2}, il = 1.0 / sqrt(g0 * g0 + g1 * g1 + g2 * g2);
                    const double amn = gv[mm / 3] * f[mm % 3] * il * cw[nn / 3] * f[nn % 3], anm = gv[nn / 3] * f[nn % 3] * il * cw[mm / 3] * f[mm % 3];
                    s_A[q][9 * mm + nn] = (ab + amn) * dO;
                    if (mm != nn) s_A[q][9 * nn + mm] = (ab + anm) * dO;
                } else {
                s_A[q][9 * mm + nn] = ab * dO; s_A[q][9 * nn + mm] = ab * dO;
                }
            }
            if (mm == nn) s_P[q][mm] = r.a * dO;
        }
        __syncthreads();
    } else {
    // A3b: the quantities shared by all entries of P and 𝔸 at a point.  256-thread workgroups: four independent parts of the block, one per wave
    // (lane = point), instead of the whole block on one lane per point while three waves wait; 64-thread workgroups: one lane per point
    {
        constexpr bool SPLIT = T >= 256;
        const int part = SPLIT ? tid >> 6 : 0, q = SPLIT ? tid & 63 : tid;
        if (q < NQ && part < 4) {
            double F[3][3];
            if constexpr (REFGRAD) { // F = I + Ĥ·J⁻¹ (every part for itself; part 0 also leaves it in the slots of s_JI, where A3c reads it)
                const double *ji = s_Ji[q], *H = s_P[q];
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int k = 0; k < 3; ++k) F[c][k] = (c == k ? 1.0 : 0.0) + H[3 * c] * ji[k] + H[3 * c + 1] * ji[3 + k] + H[3 * c + 2] * ji[6 + k];
                if (part == 0) {
#pragma unroll
                    for (int e = 0; e < 9; ++e) s_JI[q][e] = F[e / 3][e % 3];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 9; ++e) F[e / 3][e % 3] = s_JI[q][e];
            }
            HOParams mq = mat;
            if (m.act_field) { // Ta(x_q) = Tmax·Σₐ Mₐ(ξ_q)·state[cell][a] (coefficients.jl:85-99, contraction.jl:166-175)
                double ca = 0.0;
                for (int a = 0; a < 8; ++a) ca += tb.M[q][a] * m.act_field[cell * 8 + a];
                mq.Ta = mat.Ta * ca;
            }
            if (m.qp_act) { mq.Ta = m.qp_act[m.qp_stride * (cell * NQ + q)]; mq.Tb = m.qp_act[m.qp_stride * (cell * NQ + q) + 1]; }
            if (m.fsn_field) { // interpolate the nodal frame, normalise, Gram–Schmidt (microstructure.jl:176-187)
                double f[3] = {0, 0, 0}, s[3] = {0, 0, 0}, n[3] = {0, 0, 0};
                const double *fc = m.fsn_field + cell * 72;
                for (int a = 0; a < 8; ++a) {
                    const double Na = tb.M[q][a];
#pragma unroll
                    for (int d = 0; d < 3; ++d) { f[d] += Na * fc[9 * a + d]; s[d] += Na * fc[9 * a + 3 + d]; n[d] += Na * fc[9 * a + 6 + d]; }
                }
                ho_orthonormal_frame(f, s, n);
#pragma unroll
                for (int d = 0; d < 3; ++d) { mq.f[d] = f[d]; mq.s[d] = s[d]; mq.n[d] = n[d]; }
            }
            if constexpr (SPLIT) {
                if (part == 0) ho_common_part<0>(mq, F, s_C[q]);
                else if (part == 1) ho_common_part<1>(mq, F, s_C[q]);
                else if (part == 2) ho_common_part<2>(mq, F, s_C[q]);
                else ho_common_part<3>(mq, F, s_C[q]);
            } else {
                ho_common<false>(mq, F, s_C[q]);
            }
        }
    }
    __syncthreads();
    TB_MS(3);
    // A3c: one lane per (point, i, j): row (i,j) of P·dΩ and 𝔸·dΩ
    for (int t = tid; t < NQ * 9; t += T) {
        const int q = t / 9, ij = t % 9;
        double Pij;
        if constexpr (NEED_K) {
            double row[9];
            ho_row<true>(mat, s_C[q], s_JI[q], ij / 3, ij % 3, s_JI[q][9], Pij, row);
            if (m.qp_act && m.qp_stride == 5) { // rate-coupled internal variable: + (∂λ/∂F)_ij (c·w)_k f_l, the only non-symmetric part of the tangent
                const double *Fq = s_JI[q], *fv = s_C[q] + HOC_FV, *cw = m.qp_act + 5 * (cell * NQ + q) + 2;
                const int i = ij / 3, j = ij % 3;
                const double g0 = Fq[0] * fv[0] + Fq[1] * fv[1] + Fq[2] * fv[2], g1 = Fq[3] * fv[0] + Fq[4] * fv[1] + Fq[5] * fv[2],
                             g2 = Fq[6] * fv[0] + Fq[7] * fv[1] + Fq[8] * fv[2];
                const double dl = (i == 0 ? g0 : i == 1 ? g1 : g2) * fv[j] / sqrt(g0 * g0 + g1 * g1 + g2 * g2) * s_JI[q][9];
#pragma unroll
                for (int e = 0; e < 9; ++e) row[e] += dl * cw[e / 3] * fv[e % 3];
            }
#pragma unroll
            for (int e = 0; e < 9; ++e) s_A[q][9 * ij + e] = row[e];
        } else {
            ho_row<false>(mat, s_C[q], s_JI[q], ij / 3, ij % 3, s_JI[q][9], Pij, nullptr);
        }
        s_P[q][ij] = Pij;
    }
    __syncthreads();

    }
    TB_MS(4);
    double racc = 0.0;
    // rₑ[(a,c)] = Σ_q Σ_s ∂̂ₛNₐ(ξ_q) P̂_q[c][s] for the lane's element unknown (reference-gradient path; P̂ = the pulled-back stress in s_P): the reference
    // gradients of the lane's node come from its 1-D factors
    auto residual_refgrad = [&]() {
        auto PHr = [](int i, int q) constexpr { return quad1d(i, G3::x(q)); };
        auto DPr = [](int i, int q) constexpr { return dquad1d(i, G3::x(q)); };
        double acc = 0.0;
        if (tid < ND) {
            const int a = tid / 3, c = tid - 3 * a;
            double fa[3][3], da[3][3]; // [direction][point]
            const int ta = s_tix[a];
#pragma unroll
            for (int dir = 0; dir < 3; ++dir) {
                const int i = dir == 0 ? ta % 3 : dir == 1 ? (ta / 3) % 3 : ta / 9;
#pragma unroll
                for (int qq = 0; qq < 3; ++qq) {
                    fa[dir][qq] = i == 0 ? PHr(0, qq) : i == 1 ? PHr(1, qq) : PHr(2, qq);
                    da[dir][qq] = i == 0 ? DPr(0, qq) : i == 1 ? DPr(1, qq) : DPr(2, qq);
                }
            }
#pragma unroll
            for (int q2 = 0; q2 < 3; ++q2)
#pragma unroll
                for (int q1 = 0; q1 < 3; ++q1) {
                    const double m12 = fa[1][q1] * fa[2][q2], d1 = da[1][q1] * fa[2][q2], d2 = fa[1][q1] * da[2][q2];
#pragma unroll
                    for (int q0 = 0; q0 < 3; ++q0) {
                        const double *pp = s_P[(q0 + 3 * q1 + 9 * q2) % NQ] + 3 * c;
                        acc += da[0][q0] * m12 * pp[0] + fa[0][q0] * (d1 * pp[1] + d2 * pp[2]);
                    }
                }
        }
        return acc;
    };
    if constexpr (NEED_K && MFMA) {
        // B (matrix cores).  Kₑ is symmetric (𝔸 has major symmetry), so of the 6×3×2 tiles (M-tile mt, component d, N-tile nt)
        // the nine with mt ≥ 3, nt = 0 — rows (a ≥ 16, c), columns (b < 16, d): strictly below the block diagonal — are not
        // computed; the epilogue mirrors them from the tiles (mt ≤ 2, ·, nt = 1).  27 tiles on four waves (7, 7, 7, 6):
        //   A-operand kinds k = 3·mt + d;  k < 9: both N-tiles;  k ≥ 9: nt = 1 only
        //   wave 0: kinds 0, 4, 8 (both) + 9      wave 1: 1, 5 (both) + 10, 11, 12
        //   wave 2: 2, 6 (both) + 13, 14, 15      wave 3: 3, 7 (both) + 16, 17
        // Rows ≥ ND / columns ≥ NB of the padded tiles hold finite garbage that is never stored; only the K padding
        // (kk ≥ 3·NQ) must vanish, which the B operand does.
        static_assert(T == 256 && ND == 81 && NB == 27, "tile assignment below is for 6×2×3 tiles on four waves");
        constexpr int KS = (3 * NQ + 3) / 4;
        const int lane = tid & 63, wv = tid >> 6, lr = lane & 15, g = lane >> 4;
        // entries: e0, e1 → both N-tiles; e2, e3 → nt = 1; e4 → nt = 1 (waves 1, 2);  wave 0 additionally runs e2 on nt = 0
        int kind[5];
        kind[0] = wv; kind[1] = wv + 4;
        kind[2] = wv == 0 ? 8 : wv == 1 ? 10 : wv == 2 ? 13 : 16;
        kind[3] = wv == 0 ? 9 : wv == 1 ? 11 : wv == 2 ? 14 : 17;
        kind[4] = wv == 1 ? 12 : wv == 2 ? 15 : 17; // unused on waves 0 and 3
        const bool has_e4 = wv == 1 || wv == 2, e2_both = wv == 0;
        int offG[5], offA[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int mt = kind[i] / 3, d = kind[i] % 3;
            const int R = 16 * mt + lr, Rc = R < ND ? R : 0;
            offG[i] = 3 * (Rc / 3);
            offA[i] = 27 * (Rc % 3) + 3 * d;
        }
        const int b1 = 16 + lr < NB ? 16 + lr : 0;
        mfma_d4 acc[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) acc[i] = mfma_d4{0, 0, 0, 0};
        const double *sG = &s_G[0][0][0], *sA = &s_A[0][0];
#pragma unroll 3
        for (int s = 0; s < KS; ++s) {
            const int kk = 4 * s + g;
            const bool kv = kk < 3 * NQ;
            const int kc = kv ? kk : 0, q = kc / 3, l = kc - 3 * q;
            const double *gq = sG + q * (NB * 3), *aq = sA + q * 81 + l;
            double B0 = gq[3 * lr + l], B1 = gq[3 * b1 + l];
            if (!kv) { B0 = 0.0; B1 = 0.0; }
            double Aop[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const double *gg = gq + offG[i], *aa = aq + offA[i];
                Aop[i] = gg[0] * aa[0] + gg[1] * aa[9] + gg[2] * aa[18];
            }
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(Aop[0], B0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(Aop[0], B1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(Aop[1], B0, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(Aop[1], B1, acc[3], 0, 0, 0);
            acc[4] = __builtin_amdgcn_mfma_f64_16x16x4f64(Aop[2], B1, acc[4], 0, 0, 0);
            acc[5] = __builtin_amdgcn_mfma_f64_16x16x4f64(Aop[3], B1, acc[5], 0, 0, 0);
            if (e2_both) acc[6] = __builtin_amdgcn_mfma_f64_16x16x4f64(Aop[2], B0, acc[6], 0, 0, 0);
            else if (has_e4) acc[6] = __builtin_amdgcn_mfma_f64_16x16x4f64(Aop[4], B1, acc[6], 0, 0, 0);
        }
        if constexpr (NEED_R) {
            if (tid < ND) {
                for (int q = 0; q < NQ; ++q) {
                    const double *gr = s_G[q][tid / 3];
                    const double *p = s_P[q] + 3 * (tid % 3);
                    racc += gr[0] * p[0] + gr[1] * p[1] + gr[2] * p[2];
                }
            }
        }
        // C: D[row = g + 4·reg][col = lr] of tile (mt, d, nt) is Kₑ[(a,c) = 16·mt + row][(b,d), b = 16·nt + col]; tiles
        // (mt ≤ 2, nt = 1) also supply the mirrored entries Kₑ[(b,d)][(a,c)] of the tiles that were skipped
        auto put = [&](int R, int b, int d, double v) { // Kₑ[R][(b,d)]
            if (atomic == 2) { // element assembly: the stored Kₑ row
                ke[((int64_t)cell * ND + R) * ND + (KE_DMAJOR ? d * NB + b : 3 * b + d)] = v;
            } else {
                const int a = R / 3, c = R - 3 * a;
                const int64_t k = rowptr[s_dof[3 * a] + c] + blockpos[cell * (NB * NB) + a * NB + b] + d;
                if (atomic) unsafeAtomicAdd(nz + k, v); else nz[k] += v;
            }
        };
        // Symmetric-packed element matrix (rank27 given): the cell's nodes are ranked by their global dof, a 3×3 node block (a, b) is stored once, at
        // symblk(min rank, max rank), as [c][d] seen from the node of smaller rank — so the gather of a node reads, per cell, ONE contiguous run
        // (its blocks with all nodes of higher dof) and mirrors them into the rows of those nodes.  Half the bytes of the stored 81 × 81 matrix.
        // Entries go to the LDS block the operands lived in and leave as one coalesced 27 KB stream.
        auto put_sym = [&](int R, int b, int d, double v) {
            const int a = R / 3, c = R - 3 * a;
            const int ra = s_rank[a], rb = s_rank[b];
            int slot;
            if (a == b) slot = symblk(ra, ra) * 9 + 3 * c + d;
            else {
                // the entry and its transpose share a slot; the transpose is not computed when it lies in a skipped tile (row ≥ 48, column < 16)
                const bool partner_computed = !(b >= 16 && a < 16);
                if (partner_computed && ra > rb) return;
                slot = ra < rb ? symblk(ra, rb) * 9 + 3 * c + d : symblk(rb, ra) * 9 + 3 * d + c;
            }
            s_AG[slot] = v;
        };
        const bool sym = atomic == 2 && rank27 != nullptr;
        auto emit = [&](const mfma_d4 &v, int knd, int nt) {
            const int mt = knd / 3, d = knd % 3, b = 16 * nt + lr;
            if (b >= NB) return;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int R = 16 * mt + g + 4 * rg;
                if (R >= ND) continue;
                if (sym) { put_sym(R, b, d, v[rg]); continue; }
                put(R, b, d, v[rg]);
                if (nt == 1 && mt <= 2) put(3 * b + d, R / 3, R % 3, v[rg]); // transpose: row (b,d), column (a,c) = R
            }
        };
        if (sym) __syncthreads(); // every wave is done reading 𝔸 and ∇N (contraction, residual)
        emit(acc[0], kind[0], 0); emit(acc[1], kind[0], 1);
        emit(acc[2], kind[1], 0); emit(acc[3], kind[1], 1);
        emit(acc[4], kind[2], 1); emit(acc[5], kind[3], 1);
        if (e2_both) emit(acc[6], kind[2], 0);
        else if (has_e4) emit(acc[6], kind[4], 1);
        if (sym) {
            __syncthreads();
            static_assert(KE_SYM % 2 == 0 && KE_SYM <= AN + NQ * NB * 3, "staging block");
            double2 *dst = reinterpret_cast<double2 *>(ke + (int64_t)cell * KE_SYM);
            const double2 *src = reinterpret_cast<const double2 *>(s_AG);
            for (int i = tid; i < KE_SYM / 2; i += T) dst[i] = src[i];
        }
    } else if constexpr (NEED_K && SF) {
        // B (sum factorisation).  Kₑ[(a,c)][(b,d)] = Σ_q Σ_su ∂̂ₛNₐ(ξ_q) Â_q[c][s][d][u] ∂̂ᵤN_b(ξ_q) with the tangent pulled back to the reference cell,
        // Â_q[c][s][d][u] = Σ_kl J⁻¹[s][k] 𝔸_q[c][k][d][l] J⁻¹[u][l] dΩ.  Basis and Gauss rule are tensor products — node a ↔ (a₀,a₁,a₂), point
        // q ↔ (q₀,q₁,q₂), ∂̂ₛNₐ(ξ_q) = Π_dim ψ(dim == s)_{a_dim}(q_dim), ψ(false) = φ, ψ(true) = φ′ — so the sum over q is three one-dimensional
        // contractions.  After direction 1 only "s is 2 or not" (μ) still matters, so the nine (s, u) pairs collapse to four (μ, ν):
        //   stage 1  Z1[s][d][u][q₁][q₂][a₀][b₀]       = Σ_q₀ ψ(s==0)_{a₀}(q₀) ψ(u==0)_{b₀}(q₀) Â_q[c][s][d][u]
        //   stage 2  Z2[d][μ][ν][a₀][b₀][a₁][b₁][q₂]   = Σ_{s∈μ} Σ_{u∈ν} Σ_q₁ ψ(s==1)_{a₁}(q₁) ψ(u==1)_{b₁}(q₁) Z1
        //   stage 3  Kₑ[(a,c)][(b,d)]                  = Σ_μν Σ_q₂ ψ(μ)_{a₂}(q₂) ψ(ν)_{b₂}(q₂) Z2
        // per row component c: 243 lane-tasks (one per thread) for stage 1 — 36 multiply-adds out of 3 LDS reads, 9 writes — and 243 for stages 2 + 3
        // together, in registers — 81 + 90 multiply-adds out of 81 reads (no Z2 in LDS), the nine entries stored straight to Kₑ: ≈ 1.5·10⁵
        // multiply-adds per cell instead of the 1.2·10⁶ of the dense 81 × 27 × 81 products, on the vector ALUs, 53 KB of LDS (three workgroups per
        // CU), no symmetry assumed (so the rate-coupled, non-symmetric tangent runs here too).  Same sums as elements.jl:211-223 in another order.
        static_assert(T == 256 && NB == 27 && NQ == 27, "sum-factorised contraction: triquadratic field, 3×3×3 Gauss rule");
        // stage 0: 𝔸·dΩ → Â, in place, one lane per (point, c, d); the lanes d = 0 also pull the stress back, P̂[c][s] = Σ_k J⁻¹[s][k] P[c][k]·dΩ
        if (tid < NQ * 9) {
            const int q = tid / 9, cd = tid - 9 * q, c = cd / 3, d = cd - 3 * c;
            const double *ji = s_Ji[q];
            if (NEED_R && d == 0) {
                double *pp = s_P[q] + 3 * c;
                const double p0 = pp[0], p1 = pp[1], p2 = pp[2];
#pragma unroll
                for (int s_ = 0; s_ < 3; ++s_) pp[s_] = ji[3 * s_] * p0 + ji[3 * s_ + 1] * p1 + ji[3 * s_ + 2] * p2;
            }
            double *Aq = s_A[q];
            double a9[3][3], t9[3][3];
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int l = 0; l < 3; ++l) a9[k][l] = Aq[9 * (3 * c + k) + 3 * d + l];
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int u = 0; u < 3; ++u) t9[k][u] = a9[k][0] * ji[3 * u] + a9[k][1] * ji[3 * u + 1] + a9[k][2] * ji[3 * u + 2];
#pragma unroll
            for (int s_ = 0; s_ < 3; ++s_)
#pragma unroll
                for (int u = 0; u < 3; ++u) Aq[9 * (3 * c + s_) + 3 * d + u] = ji[3 * s_] * t9[0][u] + ji[3 * s_ + 1] * t9[1][u] + ji[3 * s_ + 2] * t9[2][u];
        }
        lds_barrier(); // LDS traffic only: the stores of the previous rows stay in flight
        TB_MS(5);
        double *Z1 = &s_G[0][0][0];
        auto PH = [](int i, int q) constexpr { return quad1d(i, G3::x(q)); };
        auto DP = [](int i, int q) constexpr { return dquad1d(i, G3::x(q)); };
        if constexpr (NEED_R) racc = residual_refgrad();
        // task of stages 2 + 3 (fixed for the three row components): lanes of a wave share (a₀, a₁) where they can — their stores then fall into the
        // same rows of Kₑ — and the nine lanes (a₁, b₁) of one (d, a₀, b₀) read the same Z1 words (LDS broadcast)
        int t23 = tid < 243 ? tid : 0;
        const int tb1 = t23 % 3; t23 /= 3;
        const int tb0 = t23 % 3; t23 /= 3;
        const int td = t23 % 3; t23 /= 3;
        const int ta1 = t23 % 3;
        const int ta0 = t23 / 3;
        double c2[2][2][3]; // ψ(s==1)_{a₁}(q₁) · ψ(u==1)_{b₁}(q₁)
#pragma unroll
        for (int q1 = 0; q1 < 3; ++q1) {
            const double pa = ta1 == 0 ? PH(0, q1) : ta1 == 1 ? PH(1, q1) : PH(2, q1), da = ta1 == 0 ? DP(0, q1) : ta1 == 1 ? DP(1, q1) : DP(2, q1);
            const double pb = tb1 == 0 ? PH(0, q1) : tb1 == 1 ? PH(1, q1) : PH(2, q1), db = tb1 == 0 ? DP(0, q1) : tb1 == 1 ? DP(1, q1) : DP(2, q1);
            c2[0][0][q1] = pa * pb; c2[0][1][q1] = pa * db; c2[1][0][q1] = da * pb; c2[1][1][q1] = da * db;
        }
        int rowa[3], colb[3]; // Ferrite nodes of (a₀, a₁, ·) and (b₀, b₁, ·)
#pragma unroll
        for (int k = 0; k < 3; ++k) { rowa[k] = s_node[ta0 + 3 * ta1 + 9 * k]; colb[k] = s_node[tb0 + 3 * tb1 + 9 * k]; }
        TB_MS(6);
        for (int c = 0; c < 3; ++c) {
            if (tid < 243) { // stage 1: task (s, d, u, q₁, q₂)
                int t = tid;
                const int q2 = t % 3; t /= 3;
                const int q1 = t % 3; t /= 3;
                const int u = t % 3; t /= 3;
                const int d = t % 3;
                const int s_ = t / 3;
                double in[3], tb0_[3][3];
#pragma unroll
                for (int q0 = 0; q0 < 3; ++q0) in[q0] = s_A[q0 + 3 * q1 + 9 * q2][9 * (3 * c + s_) + 3 * d + u];
#pragma unroll
                for (int b0 = 0; b0 < 3; ++b0)
#pragma unroll
                    for (int q0 = 0; q0 < 3; ++q0) tb0_[b0][q0] = (u == 0 ? DP(b0, q0) : PH(b0, q0)) * in[q0];
#pragma unroll
                for (int a0 = 0; a0 < 3; ++a0)
#pragma unroll
                    for (int b0 = 0; b0 < 3; ++b0) {
                        double v = 0.0;
#pragma unroll
                        for (int q0 = 0; q0 < 3; ++q0) v += (s_ == 0 ? DP(a0, q0) : PH(a0, q0)) * tb0_[b0][q0];
                        Z1[9 * tid + 3 * a0 + b0] = v;
                    }
            }
            lds_barrier();
            TB_MS(7 + 2 * c);
            if (tid < 243) { // stages 2 + 3 in registers: 81 LDS reads, nine entries (a₂, b₂) of row component c out
                double z[2][2][3];
#pragma unroll
                for (int e = 0; e < 12; ++e) (&z[0][0][0])[e] = 0.0;
#pragma unroll
                for (int s_ = 0; s_ < 3; ++s_)
#pragma unroll
                    for (int u = 0; u < 3; ++u)
#pragma unroll
                        for (int q1 = 0; q1 < 3; ++q1) {
                            const double cf = c2[s_ == 1][u == 1][q1];
                            const double *zp = Z1 + 9 * ((((s_ * 3 + td) * 3 + u) * 3 + q1) * 3) + 3 * ta0 + tb0;
#pragma unroll
                            for (int q2 = 0; q2 < 3; ++q2) z[s_ == 2][u == 2][q2] += cf * zp[9 * q2];
                        }
                double w[2][3][3]; // w[μ][b₂][q₂] = Σ_ν ψ(ν)_{b₂}(q₂) z[μ][ν][q₂]
#pragma unroll
                for (int mu = 0; mu < 2; ++mu)
#pragma unroll
                    for (int b2 = 0; b2 < 3; ++b2)
#pragma unroll
                        for (int q2 = 0; q2 < 3; ++q2) w[mu][b2][q2] = PH(b2, q2) * z[mu][0][q2] + DP(b2, q2) * z[mu][1][q2];
#pragma unroll
                for (int a2 = 0; a2 < 3; ++a2)
#pragma unroll
                    for (int b2 = 0; b2 < 3; ++b2) {
                        double v = 0.0;
#pragma unroll
                        for (int q2 = 0; q2 < 3; ++q2) v += PH(a2, q2) * w[0][b2][q2] + DP(a2, q2) * w[1][b2][q2];
                        // assemble!(assembler, dofs, Kₑ): entry ((a, c), (b, d))
                        if (atomic == 2) ke[((int64_t)cell * ND + 3 * (ta0 + 3 * ta1 + 9 * a2) + c) * ND + 27 * b2 + (tid < 243 ? tid : 0) % 27] = v; // tensor-order layout (tb_mech_common.hpp)
                        else {
                            const int64_t k = rowptr[s_dof[3 * rowa[a2]] + c] + blockpos[cell * (NB * NB) + rowa[a2] * NB + colb[b2]] + td;
                            if (atomic) unsafeAtomicAdd(nz + k, v); else nz[k] += v;
                        }
                    }
            }
            lds_barrier(); // Z1 is rewritten by the next component's stage 1
            TB_MS(8 + 2 * c);
        }
    } else if constexpr (REFGRAD && !NEED_K) {
        // residual only, reference-gradient path: pull the stress back (one lane per (point, c)), then contract with the reference gradients
        if (tid < NQ * 3) {
            const int q = tid / 3, c = tid - 3 * q;
            const double *ji = s_Ji[q];
            double *pp = s_P[q] + 3 * c;
            const double p0 = pp[0], p1 = pp[1], p2 = pp[2];
#pragma unroll
            for (int s_ = 0; s_ < 3; ++s_) pp[s_] = ji[3 * s_] * p0 + ji[3 * s_ + 1] * p1 + ji[3 * s_ + 2] * p2;
        }
        __syncthreads();
        racc = residual_refgrad();
    } else {
    // B: sweep the points
    const int a_own = tid / NG, bg = tid % NG;
    const bool pair_thread = tid < NB * NG;
    double Kacc[NEED_K ? PB : 1][9];
#pragma unroll
    for (int pb = 0; pb < (NEED_K ? PB : 1); ++pb)
#pragma unroll
        for (int e = 0; e < 9; ++e) Kacc[pb][e] = 0.0;
    auto stage_T = [&](int q, int buf) {
        for (int idx = tid; idx < NB * 27; idx += T) {
            const int a = idx / 27, e = idx % 27, c = e / 9, dl = e % 9; // T[a][c][d][l], dl = 3d + l
            const double *g = s_G[q][a];
            s_T[buf][a][e] = g[0] * s_A[q][9 * (3 * c + 0) + dl] + g[1] * s_A[q][9 * (3 * c + 1) + dl] + g[2] * s_A[q][9 * (3 * c + 2) + dl];
        }
    };
    if constexpr (NEED_K) { stage_T(0, 0); __syncthreads(); }
    for (int q = 0; q < NQ; ++q) {
        if constexpr (NEED_K) {
            if (q + 1 < NQ) stage_T(q + 1, (q + 1) & 1);
            if (pair_thread) {
                double Ta[27];
#pragma unroll
                for (int e = 0; e < 27; ++e) Ta[e] = s_T[q & 1][a_own][e];
#pragma unroll
                for (int pb = 0; pb < PB; ++pb) {
                    const double *gb = s_G[q][bg * PB + pb];
                    const double g0 = gb[0], g1 = gb[1], g2 = gb[2];
#pragma unroll
                    for (int cd = 0; cd < 9; ++cd) Kacc[pb][cd] += Ta[3 * cd] * g0 + Ta[3 * cd + 1] * g1 + Ta[3 * cd + 2] * g2;
                }
            }
        }
        if constexpr (NEED_R) {
            if (tid < ND) {
                const double *g = s_G[q][tid / 3];
                const double *p = s_P[q] + 3 * (tid % 3);
                racc += g[0] * p[0] + g[1] * p[1] + g[2] * p[2];
            }
        }
        if constexpr (NEED_K) __syncthreads();
    }

    // C: assemble!(assembler, dofs, Kₑ, rₑ).  The 3×3 node-pair blocks are re-dealt through LDS one component row c at a
    // time so that three consecutive lanes write the three consecutive columns (d = 0,1,2) of a block row: a
    // wave-instruction then touches 21 24-byte segments instead of 63 scattered doubles (3× fewer memory-side requests).
    if constexpr (NEED_K) {
        double *kbuf = &s_A[0][0]; // 𝔸 is dead after the sweep; NB·NB·3 ≤ NQ·81 doubles
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            __syncthreads();
            if (pair_thread) {
#pragma unroll
                for (int pb = 0; pb < PB; ++pb)
#pragma unroll
                    for (int d = 0; d < 3; ++d) kbuf[(a_own * NB + bg * PB + pb) * 3 + d] = Kacc[pb][3 * c + d];
            }
            __syncthreads();
            if (atomic == 2) { // element assembly: Kₑ rows leave as contiguous ND-long runs
                for (int idx = tid; idx < NB * NB * 3; idx += T) {
                    const int a = idx / ND, j = idx % ND;
                    ke[((int64_t)cell * ND + 3 * a + c) * ND + j] = kbuf[idx];
                }
            } else {
                for (int idx = tid; idx < NB * NB * 3; idx += T) {
                    const int d = idx % 3, ab = idx / 3, a = ab / NB;
                    const int64_t k = rowptr[s_dof[3 * a] + c] + blockpos[cell * (NB * NB) + ab] + d;
                    if (atomic) unsafeAtomicAdd(nz + k, kbuf[idx]); else nz[k] += kbuf[idx];
                }
            }
        }
    }
    }
    if constexpr (NEED_R) {
        if (tid < ND) {
            if (atomic == 2) re[cell * ND + tid] = racc;
            else if (atomic) unsafeAtomicAdd(r + s_dof[tid], racc);
            else r[s_dof[tid]] += racc;
        }
    }
#ifdef TB_ABLATION
    if (m.prof) { TB_MS(13); __builtin_amdgcn_s_waitcnt(0); TB_MS(14); }
#endif
#undef TB_MS
}


// ElementAssemblyStrategy for the tangent: second pass, one wave per node.  A lane owns *positions* of the node's three
// CSR rows, not source entries: per cell touching the node a byte map "neighbour-node slot of the row → local node b" is
// built in LDS from blockpos (27 byte stores), then every lane sums its entry over the ≤8 element matrices in cell order —
// independent loads, all in flight together, no accumulator in LDS, each nz stored once (bit-reproducible like the
// reference's EA strategy).  DMAJOR: row layout of the stored Kₑ is [d][b] instead of [b][d].
template <int NB, bool DMAJOR, int NK, bool TL = false>
__device__ __forceinline__ void gather_rows(const double *const (&rowk)[8], const uint8_t *inv, int nbr_max, int L, int lane, double *dst0, bool first)
{
    // the three rows of the node share the position → local-node lookup: one pass over the positions with the 3·NK loads of a position in flight
    // together (constant offsets c·ND from one address), instead of three passes of NK loads each
    constexpr int ND = 3 * NB;
    for (int p = lane; p < L; p += 64) {
        const int nbr = p / 3, d = p - 3 * nbr;
        double v[3][NK];
        bool ok[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int b = inv[k * nbr_max + nbr];
            ok[k] = b != 0xFF;
            const int bb = ok[k] ? b : 0;
            const double *src = rowk[k] + (TL ? bb + 9 * d : DMAJOR ? d * NB + bb : 3 * bb + d); // TL: the map holds cb(b) (tensor-order columns)
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c][k] = src[c * ND];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double sum = 0.0;
#pragma unroll
            for (int k = 0; k < NK; ++k) sum += ok[k] ? v[c][k] : 0.0;
            double *dst = dst0 + (int64_t)c * L;
            if (first) dst[p] = sum; else dst[p] += sum;
        }
    }
}

template <int NB, bool DMAJOR, bool TL = false>
__global__ void __launch_bounds__(256)
k_gather_node_rows(const int32_t *__restrict__ node_dof0, int64_t n_nodes, const int64_t *__restrict__ ea_ptr, const int32_t *__restrict__ ea_src,
                   const double *__restrict__ ke, const uint16_t *__restrict__ blockpos, const int64_t *__restrict__ rowptr, double *__restrict__ nz,
                   int nbr_max)
{
    constexpr int ND = 3 * NB, KC = 8;
    extern __shared__ uint8_t s_inv[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t node = (int64_t)blockIdx.x * (blockDim.x >> 6) + wv;
    if (node >= n_nodes) return;
    uint8_t *inv = s_inv + (size_t)wv * KC * nbr_max;
    const int32_t dof0 = node_dof0[node];
    const int64_t g0 = rowptr[dof0];
    const int L = (int)(rowptr[dof0 + 1] - g0); // rows dof0, dof0+1, dof0+2 are consecutive runs of equal length (checked on the host)
    const int64_t k0 = ea_ptr[dof0], k1 = ea_ptr[dof0 + 1];
    for (int64_t kb = k0; kb < k1; kb += KC) {
        const int nk = (int)(k1 - kb < KC ? k1 - kb : KC);
        const int nkp = nk <= 2 ? nk : nk <= 4 ? 4 : 8; // padded slots repeat cell 0 with an all-0xFF map: contribute nothing
        for (int i = lane; i < nkp * nbr_max / 4; i += 64) reinterpret_cast<uint32_t *>(inv)[i] = 0xFFFFFFFFu;
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < nk * NB; i += 64) {
            const int k = i / NB, b = i - k * NB;
            const int32_t slot = ea_src[kb + k];
            const int64_t cell = slot / ND;
            const int a = (slot % ND) / 3;
            inv[k * nbr_max + blockpos[cell * (NB * NB) + a * NB + b] / 3] = (uint8_t)(TL ? cb27(b) : b);
        }
        __builtin_amdgcn_wave_barrier();
        const double *rowk[8];
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            const int32_t slot = ea_src[kb + (k < nk ? k : 0)];
            rowk[k] = ke + ((int64_t)(slot / ND) * ND + 3 * (TL ? tix27((slot % ND) / 3) : (slot % ND) / 3)) * ND;
        }
        const bool first = kb == k0;
        if (nkp == 1) gather_rows<NB, DMAJOR, 1, TL>(rowk, inv, nbr_max, L, lane, nz + g0, first);
        else if (nkp == 2) gather_rows<NB, DMAJOR, 2, TL>(rowk, inv, nbr_max, L, lane, nz + g0, first);
        else if (nkp == 4) gather_rows<NB, DMAJOR, 4, TL>(rowk, inv, nbr_max, L, lane, nz + g0, first);
        else gather_rows<NB, DMAJOR, 8, TL>(rowk, inv, nbr_max, L, lane, nz + g0, first);
        __builtin_amdgcn_wave_barrier();
    }
}

// The same gather with the source rows staged through LDS and the per-node metadata in one record (default when a node's rows hold ≤ 384 positions
// and no node sits in more than 8 cells).  The kernel above is bound by its chain of dependent trips (dof → row pointers and slot range → slots →
// block positions → data, then one trip per 64 positions of each row): ≈ 19 µs per node with 24 waves per CU.  Here a wave reads ONE 48-byte record,
// then — in the same trip — the block positions of its cells and the cells' runs themselves (a node's share of an element matrix is one contiguous run,
// rows 3a..3a+2 of Kₑ: 3·ND doubles, read coalesced, every line once), parks both in LDS and sums from LDS into registers: a lane owns positions
// lane + 64·i of the three rows, contributions added in cell order (bit-reproducible); every nz stored once, coalesced.
struct GatherNode {
    int64_t g0;      // first nz of row dof0 (rows dof0+1, dof0+2 follow, L entries each)
    int32_t L, nk;   // row length, number of cells at the node
    int32_t slot[8]; // cell·ND + 3a: row offset of the node's run in the stored element matrices, cell-ordered
};

// LAYOUT of the stored element matrices: 0 rows / columns in Ferrite order, 1 tensor order (tb_mech_common.hpp), 2 symmetric-packed by rank (round 5): the
// 378 node blocks (i ≤ j by the rank of the nodes' dofs within the cell) of 9 doubles, block (i, j) as [c][d] seen from node i — half the bytes written by
// the contraction kernel.  A node A of rank r then finds its blocks with the nodes of higher rank in ONE contiguous run and those with the r nodes of
// lower rank as r transposed 72-byte blocks (pulled, not mirrored: the round-3 variant WROTE the mirror pieces into other rows — partial lines — and lost);
// either way 243 doubles per (node, cell) are staged, addressed by the neighbour's rank.
template <int NB, int KC, int LAYOUT = 0>
__global__ void __launch_bounds__(256, KC == 4 ? 4 : 2)
k_gather_node_rows_lds(const GatherNode *__restrict__ gn, int64_t n_nodes, const double *__restrict__ ke, const uint16_t *__restrict__ blockpos,
                       double *__restrict__ nz, int nbr_pad, const uint8_t *__restrict__ rank27)
{
    constexpr bool TL = LAYOUT == 1, SYM = LAYOUT == 2;
    constexpr int ND = 3 * NB, RUN = 3 * ND, NJ = (RUN + 63) / 64, NI = 6;
    extern __shared__ double s_stage[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t node = (int64_t)blockIdx.x * 4 + wv;
    if (node >= n_nodes) return; // no workgroup barrier below: waves are independent
    double *buf = s_stage + (size_t)wv * (KC * RUN + KC * nbr_pad / 8);
    uint8_t *inv = reinterpret_cast<uint8_t *>(buf + KC * RUN);
    const GatherNode rec = gn[node];
    const int L = rec.L;
    double acc[NI][3];
#pragma unroll
    for (int i = 0; i < NI; ++i) acc[i][0] = acc[i][1] = acc[i][2] = 0.0;
#pragma unroll
    for (int kb = 0; kb < 8; kb += KC) {
        if (kb >= rec.nk) break;
        const int nk = rec.nk - kb < KC ? rec.nk - kb : KC;
        double r[KC][NJ];
        int rA[KC]; // SYM: rank of this node in cell k (wave-uniform)
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            rA[k] = 0;
            if (k < nk) {
                int64_t row0 = rec.slot[kb + k]; // cell·ND + 3a
                if constexpr (SYM) {
                    const int64_t cellk = row0 / ND;
                    const int ra = rank27[cellk * 32 + (int)(row0 - cellk * ND) / 3];
                    rA[k] = ra;
                    const double *src = ke + cellk * KE_SYM;
                    const int diag = symblk(ra, ra);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const int idx = lane + 64 * j, rb = idx / 9, e = idx - 9 * rb;
                        const int blk = rb >= ra ? diag + (rb - ra) : symblk(rb, ra);
                        r[k][j] = idx < RUN ? src[blk * 9 + e] : 0.0;
                    }
                } else {
                if constexpr (TL) { const int64_t cellk = row0 / ND; row0 = cellk * ND + 3 * tix27((int)(row0 - cellk * ND) / 3); }
                const double *src = ke + row0 * ND;
#pragma unroll
                for (int j = 0; j < NJ; ++j) { const int idx = lane + 64 * j; r[k][j] = idx < RUN ? src[idx] : 0.0; }
                }
            }
        }
        for (int i = lane; i < nk * nbr_pad / 4; i += 64) reinterpret_cast<uint32_t *>(inv)[i] = 0xFFFFFFFFu;
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < nk * NB; i += 64) {
            const int k = i / NB, b = i - k * NB;
            int32_t slot = rec.slot[kb];
#pragma unroll
            for (int kk = 1; kk < KC; ++kk) slot = k == kk ? rec.slot[kb + kk] : slot;
            const int64_t cell = slot / ND;
            const int a = (slot - (int32_t)cell * ND) / 3;
            inv[k * nbr_pad + blockpos[cell * (NB * NB) + a * NB + b] / 3] = (uint8_t)(SYM ? rank27[cell * 32 + b] : TL ? cb27(b) : b);
        }
#pragma unroll
        for (int k = 0; k < KC; ++k)
            if (k < nk) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) { const int idx = lane + 64 * j; if (idx < RUN) buf[k * RUN + idx] = r[k][j]; }
            }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int p = lane + 64 * i;
            if (p < L) {
                const int nbr = p / 3, d = p - 3 * nbr;
#pragma unroll
                for (int k = 0; k < KC; ++k)
                    if (k < nk) {
                        const int b = inv[k * nbr_pad + nbr];
                        if (b != 0xFF) {
                            if constexpr (SYM) { // b = the neighbour's rank: block staged at b·9, as [c][d] from this node when b ≥ its rank, transposed otherwise
                                const bool up = b >= rA[k];
                                const double *sv = buf + k * RUN + 9 * b + (up ? d : 3 * d);
                                const int st = up ? 3 : 1;
                                acc[i][0] += sv[0];
                                acc[i][1] += sv[st];
                                acc[i][2] += sv[2 * st];
                            } else {
                            const double *sv = buf + k * RUN + (TL ? b + 9 * d : 3 * b + d);
                            acc[i][0] += sv[0];
                            acc[i][1] += sv[ND];
                            acc[i][2] += sv[2 * ND];
                            }
                        }
                    }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int p = lane + 64 * i;
        if (p < L) {
            nz[rec.g0 + p] = acc[i][0];
            nz[rec.g0 + L + p] = acc[i][1];
            nz[rec.g0 + 2 * (int64_t)L + p] = acc[i][2];
        }
    }
}

// Gather of the symmetric-packed element matrices (matrix-core path): one wave per node A.  Per cell around A the wave reads ONE contiguous run —
// the 3×3 blocks of A with every node of higher dof in that cell — sums the runs of its ≤ KC cells in cell order (bit-reproducible), stores the upper
// part of A's three rows coalesced, and mirrors every block into the rows of the other node B (rows (B,d), columns (A,0..2): 24-byte pieces; nine lanes
// hold one block, so a store instruction covers 21 pieces).  Every stored byte is read once and every nz written once: 14 + 14 + 19 GB at 80³ instead of
// 27 + 27 + 19.  Workgroups are dealt to the XCDs in contiguous chunks of nodes, so that the pieces of one cache line of a row — written by the waves
// of dof-consecutive nodes — meet in one L2.
template <int NB>
__global__ void __launch_bounds__(256)
k_gather_node_rows_sym(const int32_t *__restrict__ node_dof0, int64_t n_nodes, const int64_t *__restrict__ ea_ptr, const int32_t *__restrict__ ea_src,
                       const double *__restrict__ ke, const uint16_t *__restrict__ blockpos, const uint8_t *__restrict__ rank27,
                       const int64_t *__restrict__ rowptr, const int32_t *__restrict__ colidx, double *__restrict__ nz, int nbr_max, int64_t ngroups)
{
    constexpr int ND = 3 * NB, KC = 16, CH = 21; // CH neighbours × 3 components = 63 lanes per chunk
    extern __shared__ uint8_t s_raw[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t chunkg = (ngroups + 7) >> 3;
    const int64_t grp = (int64_t)(blockIdx.x & 7) * chunkg + (blockIdx.x >> 3);
    if (grp >= ngroups) return;
    const int64_t node = grp * 4 + wv;
    if (node >= n_nodes) return;
    // per-wave LDS: inv[KC][nbr_max] bytes | posAB[nbr_max] u16 | meta[KC] (cell, a, rank) | vals[63 × 3] doubles | tgt[63] int64
    const size_t per_wave = (size_t)KC * nbr_max + 2 * (size_t)nbr_max + 16 * KC + 63 * 3 * 8 + 64 * 8;
    uint8_t *base = s_raw + (size_t)wv * ((per_wave + 15) & ~(size_t)15);
    int32_t *meta = reinterpret_cast<int32_t *>(base);                       // [KC][4]: cell, a, rank of A, unused
    double *vals = reinterpret_cast<double *>(base + 16 * KC);
    long long *tgt = reinterpret_cast<long long *>(base + 16 * KC + 63 * 3 * 8);
    uint16_t *posAB = reinterpret_cast<uint16_t *>(base + 16 * KC + 63 * 3 * 8 + 64 * 8);
    uint8_t *inv = base + 16 * KC + 63 * 3 * 8 + 64 * 8 + 2 * (size_t)nbr_max;

    const int32_t dof0 = node_dof0[node];
    const int64_t g0 = rowptr[dof0];
    const int L = (int)(rowptr[dof0 + 1] - g0); // rows dof0, dof0+1, dof0+2 are consecutive runs of equal length (checked on the host)
    const int nn = L / 3;
    const int64_t k0 = ea_ptr[dof0];
    const int nk = (int)(ea_ptr[dof0 + 1] - k0); // ≤ KC (checked on the host)
    if (lane < nk) {
        const int32_t slot = ea_src[k0 + lane];
        const int32_t cell = slot / ND, a = (slot % ND) / 3;
        meta[4 * lane] = cell; meta[4 * lane + 1] = a; meta[4 * lane + 2] = rank27[(int64_t)cell * 32 + a];
    }
    for (int i = lane; i < (KC * nbr_max) / 4; i += 64) reinterpret_cast<uint32_t *>(inv)[i] = 0xFFFFFFFFu;
    __builtin_amdgcn_wave_barrier();
    const int selfpos = blockpos[(int64_t)meta[0] * (NB * NB) + meta[1] * NB + meta[1]] / 3;
    const int nu = nn - selfpos; // A itself and its neighbours of higher dof
    for (int i = lane; i < nk * NB; i += 64) {
        const int k = i / NB, b = i - k * NB;
        const int64_t cell = meta[4 * k];
        const int a = meta[4 * k + 1], ra = meta[4 * k + 2];
        const int rb = rank27[cell * 32 + b];
        if (rb >= ra) {
            const int n = blockpos[cell * (NB * NB) + a * NB + b] / 3 - selfpos;
            inv[k * nbr_max + n] = (uint8_t)(rb - ra);
            posAB[n] = blockpos[cell * (NB * NB) + b * NB + a]; // position of column (A,0) in the rows of B: the same from every cell holding both
        }
    }
    __builtin_amdgcn_wave_barrier();

    const int ln = lane / 3, d = lane - 3 * ln; // lane 63 idles
    for (int nc0 = 0; nc0 < nu; nc0 += CH) {
        const int n = nc0 + ln;
        const bool valid = lane < 63 && n < nu;
        // mirror target of this (neighbour, component): row (B, d), column (A, 0) — two dependent loads, issued before the element-matrix reads
        long long t = -1;
        if (valid && n > 0) {
            const int32_t dofB = colidx[g0 + 3 * (selfpos + n)];
            t = rowptr[dofB + d] + posAB[n];
        }
        double sum[3] = {0.0, 0.0, 0.0};
        for (int kb = 0; kb < nk; kb += 8) {
            double v[8][3];
            bool ok[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int kk = kb + k < nk ? kb + k : kb;
                const int rel = valid ? inv[kk * nbr_max + n] : 0xFF;
                ok[k] = kb + k < nk && rel != 0xFF;
                const int ra = meta[4 * kk + 2];
                const double *src = ke + (int64_t)meta[4 * kk] * KE_SYM + (symblk(ra, ra) + (ok[k] ? rel : 0)) * 9 + d;
#pragma unroll
                for (int c = 0; c < 3; ++c) v[k][c] = ok[k] ? src[3 * c] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int c = 0; c < 3; ++c) sum[c] += v[k][c];
        }
        if (valid) {
#pragma unroll
            for (int c = 0; c < 3; ++c) nz[g0 + (int64_t)c * L + 3 * (selfpos + n) + d] = sum[c];
        }
        // mirror: the block of (A, B) transposed into rows (B, d); nine consecutive lanes-slots hold one block, three consecutive ones one 24-byte piece
        if (lane < 63) {
#pragma unroll
            for (int c = 0; c < 3; ++c) vals[3 * lane + c] = sum[c];
            tgt[lane] = t;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r_ = 0; r_ < 3; ++r_) {
            const int q = 64 * r_ + lane;
            if (q < 189) {
                const long long tt = tgt[q / 3];
                if (tt >= 0) nz[tt + (q % 3)] = vals[q];
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// element residuals → global residual: r[d] = Σ (in cell order) rₑ slots of dof d
__global__ void k_gather_residual(const int64_t *__restrict__ ptr, const int32_t *__restrict__ src, const double *__restrict__ re, int64_t ndofs,
                                  double *__restrict__ r)
{
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= ndofs) return;
    double s = 0.0;
    for (int64_t k = ptr[d]; k < ptr[d + 1]; ++k) s += re[src[k]];
    r[d] = s;
}

// ------------------------------------------------------------------------------------------------ host side
static HOParams make_params(const tb_form *f)
{
    HOParams p{};
    const double *q = f->mat.p;
    p.a = q[0]; p.b = q[1]; p.af = q[2]; p.bf = q[3]; p.as = q[4]; p.bs = q[5]; p.afs = q[6]; p.bfs = q[7]; p.beta = q[8];
    for (int i = 0; i < 3; ++i) { p.f[i] = f->mat.f[i]; p.s[i] = f->mat.s[i]; p.n[i] = f->mat.n[i]; }
    p.Ta = f->cond_model ? 0.0 : f->act_tension; // condensed: the per-point (a, b) replace the uniform tension
    p.Tb = 0.0;
    return p;
}

static bool material_is_fast_path(const tb_material &mat) { return mat.kind == TB_MATERIAL_HOLZAPFEL_OGDEN_2009 && mat.reserved == PEN_SIMPLE; }
static bool form_is_fast_path(const tb_form *f) { return material_is_fast_path(f->mat) && f->hill == 0 && !f->prestressed; }

static EnergyParams make_energy_params(const tb_form *f)
{
    EnergyParams e{};
    e.energy = f->mat.kind; e.penalty = f->mat.reserved;
    for (int i = 0; i < 9; ++i) e.p[i] = f->mat.p[i];
    for (int i = 0; i < 3; ++i) e.u[i] = f->mat.p[10 + i];
    e.Ta = f->cond_model ? 0.0 : f->act_tension;
    e.hill = f->hill; e.act_energy = f->act_energy; e.act_penalty = f->act_penalty; e.adg = f->adg; e.sarc = f->sarc;
    for (int i = 0; i < 9; ++i) e.ap[i] = f->act_p[i];
    for (int i = 0; i < 3; ++i) e.au[i] = f->act_p[9 + i];
    e.kappa = f->hill_kappa; e.sp[0] = f->sarc_p[0]; e.sp[1] = f->sarc_p[1];
    e.prestressed = f->prestressed;
    for (int i = 0; i < 9; ++i) e.G[i] = f->prestress_G[i];
    return e;
}

int host_material_eval(const tb_material *mat, const double *F9, double *psi, double *P, double *A)
{
    tb_form tmpf;
    tmpf.mat = *mat;
    return host_material_eval_form(&tmpf, F9, psi, P, A);
}

int host_material_eval_form(tb_form *form, const double *F9, double *psi, double *P, double *A)
{
    const tb_material *mat = &form->mat;
    form->act_tension = mat->p[9];
    if (!form_is_fast_path(form)) { // any energy of tb_energy.hpp: the same hyper-dual evaluation the kernels run, on the host
        const EnergyParams e = make_energy_params(form);
        const double f0[3] = {mat->f[0], mat->f[1], mat->f[2]}, s0[3] = {mat->s[0], mat->s[1], mat->s[2]}, n0[3] = {mat->n[0], mat->n[1], mat->n[2]};
        for (int pr = 0; pr < 45; ++pr) {
            int mm, nn;
            pair_components(pr, mm, nn);
            const HD r = energy_pair(e, F9, mm, nn, f0, s0, n0, e.Ta);
            if (psi) *psi = r.v;
            if (P && mm == nn) P[mm] = r.a;
            if (A) { A[9 * mm + nn] = r.ab; A[9 * nn + mm] = r.ab; }
        }
        return TB_OK;
    }
    const HOParams p = make_params(form);
    double F[3][3], Pl[9], Al[81];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) F[i][j] = F9[3 * i + j];
    // evaluated through the same split routines the kernels use (ho_common + ho_row)
    double C[HOC_SIZE];
    ho_common(p, F, C);
    for (int ij = 0; ij < 9; ++ij) ho_row<true>(p, C, F9, ij / 3, ij % 3, 1.0, Pl[ij], Al + 9 * ij);
    const double v = C[HOC_PSI];
    if (psi) *psi = v;
    if (P) for (int i = 0; i < 9; ++i) P[i] = Pl[i];
    if (A) for (int i = 0; i < 81; ++i) A[i] = Al[i];
    return TB_OK;
}

// first-component dof of every node of the vector field (rows of a node are dof0, dof0+1, dof0+2)
static int build_node_list(tb_mesh *m)
{
    std::vector<int32_t> d0;
    d0.reserve((size_t)m->ndofs / 3);
    std::vector<uint8_t> seen(m->ndofs, 0);
    for (int64_t i = 0; i < m->n_cells * m->nb; ++i) {
        const int32_t d = m->h_cell_dofs[3 * i];
        if (!seen[d]) { seen[d] = 1; d0.push_back(d); }
    }
    std::sort(d0.begin(), d0.end());
    m->n_nodes_field = (int64_t)d0.size();
    m->h_node_dof0 = d0;
    return upload(m->dev, d0, &m->d_node_dof0);
}

// the gather kernel relies on the three rows of a node being equal-length runs of whole nodes
static int check_node_rows(tb_pattern *p)
{
    if (p->max_row_len) return TB_OK;
    const tb_mesh *m = p->mesh;
    for (int32_t d : m->h_node_dof0) {
        const int64_t L = p->h_rowptr[d + 1] - p->h_rowptr[d];
        if (L % 3 || L != p->h_rowptr[d + 2] - p->h_rowptr[d + 1] || L != p->h_rowptr[d + 3] - p->h_rowptr[d + 2] || L / 3 > 254) {
            set_error("element assembly gather: rows of node dof %d are not three equal runs of whole nodes (L = %lld)", d, (long long)L);
            return TB_ERR_PATTERN;
        }
        p->max_row_len = std::max<int64_t>(p->max_row_len, L);
    }
    return TB_OK;
}

// node records of the staged gather (GatherNode): built once per pattern from the host dof table, cells in ascending order like the slot lists
static int ensure_gather_nodes(tb_pattern *p)
{
    if (p->gnodes_state) return TB_OK;
    TB_NO_CAPTURE(p->mesh->dev);
    const tb_mesh *m = p->mesh;
    const int nb = m->nb, nd = 3 * nb;
    const int64_t nn = m->n_nodes_field;
    if (p->max_row_len > 384 || m->n_cells * (int64_t)nd >= (int64_t)0x7fffffff) { p->gnodes_state = -1; return TB_OK; }
    std::vector<int32_t> node_of((size_t)m->ndofs, -1);
    for (int64_t i = 0; i < nn; ++i) node_of[m->h_node_dof0[i]] = (int32_t)i;
    std::vector<GatherNode> recs((size_t)nn);
    for (int64_t i = 0; i < nn; ++i) {
        const int32_t d = m->h_node_dof0[i];
        recs[i].g0 = p->h_rowptr[d];
        recs[i].L = (int32_t)(p->h_rowptr[d + 1] - p->h_rowptr[d]);
        recs[i].nk = 0;
        for (int k = 0; k < 8; ++k) recs[i].slot[k] = 0;
    }
    for (int64_t c = 0; c < m->n_cells; ++c)
        for (int a = 0; a < nb; ++a) {
            GatherNode &g = recs[node_of[m->h_cell_dofs[c * nd + 3 * a]]];
            if (g.nk >= 8) { p->gnodes_state = -1; return TB_OK; }
            g.slot[g.nk++] = (int32_t)(c * nd + 3 * a);
        }
    // running maximum of the last contributing cell over the records (dof order): the records before the first one whose value reaches c are complete
    // once cells [0, c) are integrated — the prefix the chunked linearisation gathers behind the integration front.  On a mesh numbered cell by cell
    // the prefix trails the front by about one layer of cells.
    p->h_gn_last.resize((size_t)nn);
    int32_t run = 0;
    for (int64_t i = 0; i < nn; ++i) { run = std::max(run, recs[i].slot[recs[i].nk - 1] / nd); p->h_gn_last[i] = run; }
    TB_HIP(hipMalloc(&p->d_gnodes, sizeof(GatherNode) * (size_t)nn));
    TB_HIP(hipMemcpy(p->d_gnodes, recs.data(), sizeof(GatherNode) * (size_t)nn, hipMemcpyHostToDevice));
    p->gnodes_state = 1;
    return TB_OK;
}

// Per cell: rank of every field node among the cell's nodes by global dof (27 bytes, padded to 32) — the order in which the symmetric-packed
// element matrix stores its node blocks.  rank27_ok < 0: some node sits in more than 16 cells (the mirroring gather keeps 16 cell maps per wave).
static int ensure_rank27(tb_mesh *m)
{
    if (m->rank27_ok) return TB_OK;
    const int nb = m->nb;
    if (nb != 27) { m->rank27_ok = -1; return TB_OK; }
    std::vector<uint8_t> rk((size_t)m->n_cells * 32, 0);
    std::vector<uint8_t> cnt((size_t)m->ndofs, 0);
    bool ok = true;
#pragma omp parallel for schedule(static)
    for (int64_t c = 0; c < m->n_cells; ++c) {
        const int32_t *d = &m->h_cell_dofs[c * 3 * nb];
        for (int a = 0; a < nb; ++a) {
            int r = 0;
            for (int b = 0; b < nb; ++b) r += d[3 * b] < d[3 * a];
            rk[(size_t)c * 32 + a] = (uint8_t)r;
        }
    }
    for (int64_t i = 0; i < m->n_cells * nb; ++i) { uint8_t &k = cnt[m->h_cell_dofs[3 * i]]; if (k == 255 || ++k > 16) ok = false; }
    if (!ok) { m->rank27_ok = -1; return TB_OK; }
    int rc = upload(m->dev, rk, &m->d_rank27);
    if (rc) return rc;
    m->rank27_ok = 1;
    return TB_OK;
}

int ensure_blockpos(tb_pattern *p)
{
    if (p->d_blockpos) return TB_OK;
    tb_mesh *m = p->mesh;
    const int64_t n = m->n_cells * m->nb * m->nb;
    TB_HIP(hipMalloc((void **)&p->d_blockpos, sizeof(uint16_t) * n));
    hipLaunchKernelGGL(k_build_blockpos, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, m->dev->stream, m->d_cell_dofs, m->n_cells, m->nb,
                       p->d_rowptr, p->d_colidx, p->d_blockpos, m->dev->d_status);
    TB_HIP(hipGetLastError());
    return TB_OK;
}

template <class FE, bool NEED_K, bool NEED_R, int CT, bool AD>
static int run(tb_form *f, tb_pattern *p, int strategy, const double *d_u, double *d_nz, double *d_r)
{
    constexpr bool MFMA = CT == 1;
    tb_mesh *m = f->mesh;
    tb_device *dev = m->dev;
    constexpr bool refgrad_r = !NEED_K && NEED_R && FE::NB == 27 && FE::NQ == 27 && FE::THREADS == 256; // residual-only kernel on the reference-gradient path
    if ((NEED_K && CT == 2) || refgrad_r) { int rcx = ensure_cell_xyz(m); if (rcx) return rcx; }
    MechMesh mm{m->d_xyz, m->d_conn, m->d_cell_dofs, f->d_field, f->cond_model ? nullptr : f->d_act_field, f->cond_model ? f->d_qp_act : nullptr, f->d_u_prev ? 5 : 2,
                m->d_cell_xyz};
#ifdef TB_ABLATION
    static long long *d_mprof = nullptr;
    const int nmprof = (int)(m->n_cells >> 8) + 1;
    if (!d_mprof && getenv("TB_PROF_STAMPS") && NEED_K) TB_HIP(hipMalloc((void **)&d_mprof, (size_t)nmprof * 16 * sizeof(long long)));
    if (d_mprof && NEED_K) TB_HIP(hipMemsetAsync(d_mprof, 0, (size_t)nmprof * 16 * sizeof(long long), dev->stream));
    mm.prof = NEED_K ? d_mprof : nullptr;
#endif
    const HOParams hp = make_params(f);
    const EnergyParams ep = make_energy_params(f);
    const bool ea = strategy == TB_STRATEGY_ELEMENT || strategy == TB_STRATEGY_PATCH;
    if (ea && (f->has_cellset || f->accumulate)) {
        set_error("hyperelastic assembly: subdomain / accumulating forms need TB_STRATEGY_PER_COLOR or TB_STRATEGY_ATOMIC (the element strategy writes whole rows)");
        return TB_ERR_UNSUPPORTED;
    }
    if (NEED_K) {
        int rc = ensure_blockpos(p);
        if (rc) return rc;
        if (!ea && !f->accumulate) TB_HIP(hipMemsetAsync(d_nz, 0, (size_t)p->nnz * sizeof(double), dev->stream));
    }
    if (NEED_R && !ea && !f->accumulate) TB_HIP(hipMemsetAsync(d_r, 0, (size_t)m->ndofs * sizeof(double), dev->stream));
    const int64_t *rowptr = p ? p->d_rowptr : nullptr;
    const uint16_t *bp = p ? p->d_blockpos : nullptr;
    double *kebuf = nullptr, *rebuf = nullptr;
    const uint8_t *rank27 = nullptr;
    constexpr size_t dyn_lds = sizeof(double) * ((NEED_K ? FE::NQ : 1) * 81 + (refgrad_r ? 0 : FE::NQ * FE::NB * 3)); // 𝔸·dΩ + mapped gradients (s_AG; no gradient block on the reference-gradient residual path)
    auto go = [&](const int32_t *list, int64_t n, int atomic) -> int {
        if (!n) return TB_OK;
        auto kern = k_hyperelastic<FE, NEED_K, NEED_R, CT, AD>;
        TB_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)n), dim3(FE::THREADS), dyn_lds, dev->stream, mm, hp, ep, list, d_u, d_nz, d_r,
                           rowptr, bp, atomic, kebuf, rebuf, dev->d_status, rank27);
        TB_HIP(hipGetLastError());
        return TB_OK;
    };
    if (ea) {
        // ElementAssemblyStrategy (default for mechanics): Kₑ / rₑ stored per cell, then gathered per node row
        if (!m->ea) { int rc = build_ea_plan(m); if (rc) return rc; }
        rebuf = m->ea->d_ea;
        int rc;
        // TB_MECH_KE=sym: symmetric-packed element matrices + mirroring gather on the matrix-core path (its tangent is symmetric by construction; meshes
        // where no node sits in more than 16 cells).  A measured alternative, parity-green, slower on MI355X than the full 81 × 81 storage although it
        // moves 47 GB instead of 73 GB at 80³: integration 23.2 vs 21.3 ms (the store was never the bound; the LDS staging adds two barriers),
        // gather 15.2 vs 13.3 ms (24-byte mirror pieces cost more than the halved reads save) — profiles/r03_v1/mechanics_sym_vs_full.txt
        bool sym = false;
        if (NEED_K && MFMA) {
            const char *e = tune_env("TB_MECH_KE");
            if (e && !strcmp(e, "sym")) { rc = ensure_rank27(m); if (rc) return rc; sym = m->rank27_ok > 0; }
        }
        // split linearisation (round 5; comment above k_mech_points in tb_mech_split.hip): point kernel + contraction kernel instead of the fused one —
        // every symmetric tangent of the triquadratic field (the rate-coupled internal variable adds a non-symmetric term: fused kernel)
        bool split = false, ke_sym2 = false;
        if constexpr (NEED_K && CT == 2 && FE::NB == 27) {
            const char *e = tune_env("TB_MECH_SPLIT");
            split = !(e && atoi(e) == 0) && !(f->cond_model && f->d_u_prev);
            // TB_MECH_KE=sym2 (profiling build only): the element matrices symmetric-packed by rank — half the bytes of the largest stream — with a staged gather
            // that PULLS the transposed 72-byte blocks of the lower-ranked neighbours.  Parity-green and SLOWER (round 5, 80³, same box): the contraction
            // kernel is unchanged (9.0 ms: it was never bound by its stores), the gather takes 13.1 instead of 9.4 ms — a wave-load that touches seven
            // separate 72-byte blocks costs the address path more than the halved bytes save — linearize 23.6–24.0 against 19.6–19.9 ms.  Third negative
            // for symmetric storage (round 3: mirror writes from the matrix-core kernel): the 81 × 81 tensor-order form stays.
            if (split && tune_env("TB_MECH_KE") && !strcmp(tune_env("TB_MECH_KE"), "sym2")) {
                if (!m->d_node_dof0) { rc = build_node_list(m); if (rc) return rc; }
                rc = check_node_rows(p);
                if (rc) return rc;
                rc = ensure_gather_nodes(p);
                if (rc) return rc;
                rc = ensure_rank27(m);
                if (rc) return rc;
                ke_sym2 = p->gnodes_state > 0 && m->rank27_ok > 0;
            }
        }
        if (NEED_K) {
            const size_t need = sizeof(double) * (size_t)m->n_cells * (sym || ke_sym2 ? (size_t)KE_SYM : (size_t)FE::ND * FE::ND);
            if (p->d_kebuf && p->kebuf_bytes < need) { (void)hipFree(p->d_kebuf); p->d_kebuf = nullptr; }
            if (!p->d_kebuf) {
                hipError_t e = hipMalloc((void **)&p->d_kebuf, need);
                if (e != hipSuccess) { set_error("element-matrix buffer (%zu B): %s", need, hipGetErrorString(e)); return TB_ERR_NOMEM; }
                p->kebuf_bytes = need;
            }
            kebuf = p->d_kebuf;
            if (sym) rank27 = m->d_rank27;
        }
        // Chunked linearisation (TB_MECH_CHUNKS=n; default 8 for large Q2 meshes, see below; 0 / 1 = one launch): the cells go in n launches, and behind each the
        // staged gather of the node rows that chunk completes runs on a second queue — the HBM-bound gather (46 GB at 80³) beside the LDS / VALU-bound
        // integration of the next chunk.  Same kernels, same sums (a node's cells are still added in cell order): the unchunked result bit for bit.
        // Measured at 80³ (kernel trace, profiles/r04_v2/mechanics_chunk_timeline.txt): the pairs do run side by side, but a CU that holds three
        // integration workgroups has neither LDS (3 × 53 KB) nor registers (3 × 168) left, so every gather workgroup displaces an integration one —
        // 2.55 ms per pair against 1.42 + 1.18 ms alone; the gain is what the pipeline ends and the dispatch gaps cost before: 21.3 → 20.7 ms.
        // default: the triquadratic field from 262 144 cells (below that, and for the first-order field, sixteen small launches cost more than the overlap returns:
        // the 111 616-cell ventricle with a Q1 displacement took 1.26 instead of 1.11 ms)
        const int chunks_default = FE::NB == 27 && m->n_cells >= 262144 ? 8 : 0;
        const int chunks_env = [&] { const char *e = getenv("TB_MECH_CHUNKS"); return e ? atoi(e) : chunks_default; }(); // read per call: a 20 ms operation
        auto go_ea = [&](int64_t c0, int64_t n) -> int { // cells [c0, c0 + n) → stored Kₑ / rₑ
            if constexpr (NEED_K && CT == 2 && FE::NB == 27) {
                if (split) {
                    if (!n) return TB_OK;
                    mm.cell0 = c0;
                    int rcs = launch_mech_points(dev, mm, hp, AD ? &ep : nullptr, d_u, n, p->d_qpbuf);
                    if (!rcs) rcs = launch_mech_contract(dev, p->d_qpbuf, c0, n, kebuf, NEED_R ? rebuf : nullptr, ke_sym2 ? m->d_rank27 : nullptr);
                    if (rcs) return rcs;
                    mm.cell0 = 0;
                    return TB_OK;
                }
            }
            mm.cell0 = c0;
            const int rcg = go(nullptr, n, 2);
            mm.cell0 = 0;
            return rcg;
        };
        auto ensure_qpbuf = [&](int64_t ncell_max) -> int {
            const size_t need = sizeof(double) * (size_t)ncell_max * QP_REC;
            if (p->d_qpbuf && p->qpbuf_bytes < need) { (void)hipFree(p->d_qpbuf); p->d_qpbuf = nullptr; }
            if (!p->d_qpbuf) {
                hipError_t e = hipMalloc((void **)&p->d_qpbuf, need);
                if (e != hipSuccess) { (void)hipGetLastError(); set_error("quadrature-point record buffer (%zu B): %s", need, hipGetErrorString(e)); return TB_ERR_NOMEM; }
                p->qpbuf_bytes = need;
            }
            return TB_OK;
        };
        bool chunked = false;
        if (NEED_K && !sym && !MFMA && chunks_env > 1 && m->n_cells >= 4096 * (int64_t)chunks_env) {
            if (!m->d_node_dof0) { rc = build_node_list(m); if (rc) return rc; }
            rc = check_node_rows(p);
            if (rc) return rc;
            rc = ensure_gather_nodes(p);
            if (rc) return rc;
            chunked = p->gnodes_state > 0;
        }
        if (chunked) {
            rc = ensure_aux_stream(dev);
            if (rc) return rc;
            const int nbr_pad = ((((int)p->max_row_len / 3 + 3) & ~3) + 7) & ~7;
            const size_t glds = (size_t)4 * 4 * (3 * FE::ND * sizeof(double) + (size_t)nbr_pad);
            int64_t n_done = 0;
            if (split) { rc = ensure_qpbuf(m->n_cells / chunks_env + 1); if (rc) return rc; }
            const int rcl = [&]() -> int {
                for (int k = 0; k < chunks_env; ++k) {
                    const int64_t c0 = m->n_cells * k / chunks_env, c1 = m->n_cells * (k + 1) / chunks_env;
                    rc = go_ea(c0, c1 - c0);
                    if (rc) return rc;
                    TB_HIP(hipEventRecord(dev->aux_ev[0], dev->stream));
                    TB_HIP(hipStreamWaitEvent(dev->aux_stream, dev->aux_ev[0], 0));
                    const int64_t n1 = k + 1 == chunks_env ? m->n_nodes_field : std::upper_bound(p->h_gn_last.begin(), p->h_gn_last.end(), (int32_t)(c1 - 1)) - p->h_gn_last.begin();
                    if (n1 > n_done) {
                        const auto launch_g = [&](auto kg) { hipLaunchKernelGGL(kg, dim3((unsigned)((n1 - n_done + 3) / 4)), dim3(256), glds, dev->aux_stream, (const GatherNode *)p->d_gnodes + n_done,
                                           n1 - n_done, kebuf, bp, d_nz, nbr_pad, ke_sym2 ? m->d_rank27 : nullptr); };
                        TB_IF_ABLATION(if (ke_sym2) launch_g(k_gather_node_rows_lds<FE::NB, 4, 2>); else) launch_g(k_gather_node_rows_lds<FE::NB, 4, (NEED_K && CT == 2 && FE::NB == 27) ? 1 : 0>);
                        TB_HIP(hipGetLastError());
                        n_done = n1;
                    }
                }
                return TB_OK;
            }();
            // the second queue is joined on every path, also behind an error half way through the chunks
            TB_HIP(hipEventRecord(dev->aux_ev[1], dev->aux_stream));
            TB_HIP(hipStreamWaitEvent(dev->stream, dev->aux_ev[1], 0));
            if (rcl) return rcl;
        } else {
        if (split) { rc = ensure_qpbuf(m->n_cells); if (rc) return rc; }
        rc = go_ea(0, m->n_cells);
        if (rc) return rc;
        }
#ifdef TB_ABLATION
        if (mm.prof) { // average phase durations of the sampled workgroups (µs; wall clock 100 MHz)
            std::vector<long long> h((size_t)nmprof * 16);
            TB_SYNC_STREAM(dev);
            TB_HIP(hipMemcpy(h.data(), mm.prof, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
            double ph[14] = {0};
            int cnt = 0;
            for (int w = 0; w < nmprof; ++w) {
                const long long *a = &h[(size_t)w * 16];
                if (!a[0] || !a[14]) continue;
                for (int k = 0; k < 14; ++k) ph[k] += (double)(a[k + 1] - a[k]) * 0.01;
                ++cnt;
            }
            if (cnt) {
                fprintf(stderr, "[tbhip] mechanics phases (us, n=%d): load %.2f | A1+H %.2f | F+common %.2f | rows %.2f | stage0 %.2f | racc+setup %.2f |", cnt,
                        ph[0] / cnt, ph[1] / cnt, ph[2] / cnt, ph[3] / cnt, ph[4] / cnt, ph[5] / cnt);
                for (int c = 0; c < 3; ++c) fprintf(stderr, " c%d: s1 %.2f s23 %.2f |", c, ph[6 + 2 * c] / cnt, ph[7 + 2 * c] / cnt);
                fprintf(stderr, " tail %.2f | drain %.2f\n", ph[12] / cnt, ph[13] / cnt);
            }
        }
#endif
        if (NEED_K && sym) {
            if (!m->d_node_dof0) { rc = build_node_list(m); if (rc) return rc; }
            rc = check_node_rows(p);
            if (rc) return rc;
            const int nbr_max = (((int)p->max_row_len / 3) + 3) & ~3;
            const size_t per_wave = ((size_t)16 * nbr_max + 2 * (size_t)nbr_max + 16 * 16 + 63 * 3 * 8 + 64 * 8 + 15) & ~(size_t)15;
            const int64_t ngroups = (m->n_nodes_field + 3) / 4;
            auto k = k_gather_node_rows_sym<FE::NB>;
            hipLaunchKernelGGL(k, dim3((unsigned)(8 * ((ngroups + 7) / 8))), dim3(256), 4 * per_wave, dev->stream, m->d_node_dof0, m->n_nodes_field, m->ea->d_ptr,
                               m->ea->d_src, kebuf, bp, rank27, rowptr, p->d_colidx, d_nz, nbr_max, ngroups);
            TB_HIP(hipGetLastError());
        } else if (NEED_K && !chunked) {
            if (!m->d_node_dof0) { rc = build_node_list(m); if (rc) return rc; }
            rc = check_node_rows(p);
            if (rc) return rc;
            const int nbr_max = (((int)p->max_row_len / 3) + 3) & ~3;
            // TB_MECH_GATHER=direct: the trip-bound kernel above, kept as the comparison build (12.8 ms against 9.1 ms at 80³; a variant staging all 8
            // cells of a vertex node at once was slower still: 66 KB of LDS per workgroup leave 8 waves per CU)
            static const bool direct = [] { const char *e = tune_env("TB_MECH_GATHER"); return e && !strcmp(e, "direct"); }();
            if (!(MFMA && KE_DMAJOR) && !direct) { rc = ensure_gather_nodes(p); if (rc) return rc; }
            if (!(MFMA && KE_DMAJOR) && !direct && p->gnodes_state > 0) {
                const int nbr_pad = (nbr_max + 7) & ~7;
                const size_t lds = (size_t)4 * 4 * (3 * FE::ND * sizeof(double) + (size_t)nbr_pad);
                const auto launch_g = [&](auto k) { hipLaunchKernelGGL(k, dim3((unsigned)((m->n_nodes_field + 3) / 4)), dim3(256), lds, dev->stream, (const GatherNode *)p->d_gnodes,
                                   m->n_nodes_field, kebuf, bp, d_nz, nbr_pad, ke_sym2 ? m->d_rank27 : nullptr); };
                TB_IF_ABLATION(if (ke_sym2) launch_g(k_gather_node_rows_lds<FE::NB, 4, 2>); else) launch_g(k_gather_node_rows_lds<FE::NB, 4, (NEED_K && CT == 2 && FE::NB == 27) ? 1 : 0>);
            } else {
                const size_t lds = (size_t)4 * 8 * nbr_max;
                auto k = k_gather_node_rows<FE::NB, MFMA && KE_DMAJOR, (NEED_K && CT == 2 && FE::NB == 27)>;
                hipLaunchKernelGGL(k, dim3((unsigned)((m->n_nodes_field + 3) / 4)), dim3(256), lds, dev->stream, m->d_node_dof0, m->n_nodes_field,
                                   m->ea->d_ptr, m->ea->d_src, kebuf, bp, rowptr, d_nz, nbr_max);
            }
            TB_HIP(hipGetLastError());
        }
        if (NEED_R) {
            hipLaunchKernelGGL(k_gather_residual, dim3((unsigned)((m->ndofs + 255) / 256)), dim3(256), 0, dev->stream, m->ea->d_ptr, m->ea->d_src,
                               rebuf, m->ndofs, d_r);
            TB_HIP(hipGetLastError());
        }
        return TB_OK;
    }
    if (strategy == TB_STRATEGY_PER_COLOR) {
        const ColorPlan *cp;
        if (f->has_cellset) {
            if (!f->set_colors) { int rc = build_color_plan_subset(m, f->h_cellset, f->set_colors); if (rc) return rc; }
            cp = f->set_colors.get();
        } else {
            if (!m->colors) { int rc = build_color_plan(m); if (rc) return rc; }
            cp = m->colors.get();
        }
        for (int c = 0; c < cp->ncolors; ++c) {
            int rc = go(cp->d_cells + cp->offsets[c], cp->offsets[c + 1] - cp->offsets[c], 0);
            if (rc) return rc;
        }
        return TB_OK;
    }
    if (strategy == TB_STRATEGY_ATOMIC) return f->has_cellset ? go(f->d_cellset, f->n_set, 1) : go(nullptr, m->n_cells, 1);
    set_error("hyperelastic assembly: unknown strategy %d", strategy);
    return TB_ERR_UNSUPPORTED;
}

// ---- condensed internal variables, stage 0: fibre stretch λ = ‖F f₀‖ and calcium at every quadrature point (the inputs of the local
// problem; solve_local_constraint, materials.jl:1575-1590).  One workgroup per cell, one lane per quadrature point.
template <class FE>
__global__ void __launch_bounds__(64)
k_fiber_stretch(MechMesh m, HOParams mat, const double *__restrict__ act_field, double act_scale, const double *__restrict__ u, double *__restrict__ lam,
                double *__restrict__ ca, const double *__restrict__ u_prev, double inv_dt, double *__restrict__ vel, double *__restrict__ wout, Status *st,
                const int32_t *__restrict__ list)
{
    constexpr int NB = FE::NB, NQ = FE::NQ, ND = FE::ND;
    const MechTables<FE> &tb = g_mech_tables<FE>;
    const int64_t cell = list ? list[blockIdx.x] : blockIdx.x;
    const int tid = threadIdx.x;
    __shared__ double s_ue[ND], s_x[24], s_up[ND];
    for (int i = tid; i < ND; i += 64) { const int32_t d = m.cell_dofs[cell * ND + i]; s_ue[i] = u[d]; s_up[i] = u_prev ? u_prev[d] : 0.0; }
    for (int i = tid; i < 24; i += 64) s_x[i] = m.xyz[3 * (int64_t)m.conn[cell * 8 + i / 3] + i % 3];
    __syncthreads();
    if (tid >= NQ) return;
    const int q = tid;
    double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int k = 0; k < 3; ++k) J[i][k] += s_x[3 * a + i] * tb.dM[q][a][k];
    const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1], c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2], c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
    const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02, id = 1.0 / det;
    if (!(det > 0.0)) { st->neg_detj = 1; st->cell = cell; }
    const double ji[9] = {c00 * id, (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id, (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * id,
                          c01 * id, (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id, (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id,
                          c02 * id, (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * id, (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * id};
    double f[3] = {mat.f[0], mat.f[1], mat.f[2]};
    if (m.fsn_field) {
        f[0] = f[1] = f[2] = 0.0;
        const double *fc = m.fsn_field + cell * 72;
        for (int a = 0; a < 8; ++a) for (int d = 0; d < 3; ++d) f[d] += tb.M[q][a] * fc[9 * a + d];
        const double nf = sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
        for (int d = 0; d < 3; ++d) f[d] /= nf;
    }
    // F f₀ = f₀ + Σₐ uₐ (∇Nₐ · f₀)
    double g[3] = {f[0], f[1], f[2]}, gp[3] = {f[0], f[1], f[2]};
    for (int a = 0; a < NB; ++a) {
        const double d0 = tb.dN[q][a][0], d1 = tb.dN[q][a][1], d2 = tb.dN[q][a][2];
        double gf = 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) gf += (d0 * ji[k] + d1 * ji[3 + k] + d2 * ji[6 + k]) * f[k];
#pragma unroll
        for (int c = 0; c < 3; ++c) { g[c] += s_ue[3 * a + c] * gf; gp[c] += s_up[3 * a + c] * gf; }
    }
    double c = act_scale;
    if (act_field) { double s2 = 0.0; for (int a = 0; a < 8; ++a) s2 += tb.M[q][a] * act_field[cell * 8 + a]; c *= s2; }
    const double l = sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
    lam[cell * NQ + q] = l;
    ca[cell * NQ + q] = c;
    if (u_prev) { // rate-coupled: Ḟ f₀ = (F − F_prev) f₀ / Δt; dλ/dt = ∂λ/∂F : Ḟ = g·ġ/λ; ∂²λ/∂F² : Ḟ = w ⊗ f₀, w = ġ/λ − g (g·ġ)/λ³
        const double gd[3] = {(g[0] - gp[0]) * inv_dt, (g[1] - gp[1]) * inv_dt, (g[2] - gp[2]) * inv_dt};
        const double ggd = g[0] * gd[0] + g[1] * gd[1] + g[2] * gd[2];
        vel[cell * NQ + q] = ggd / l;
#pragma unroll
        for (int d = 0; d < 3; ++d) wout[3 * (cell * NQ + q) + d] = gd[d] / l - g[d] * ggd / (l * l * l);
    }
}

// (a, b, c) of the local solve and w of stage 0 → the (a, b, c·w) records the element kernels read
__global__ void __launch_bounds__(256) k_pack_rate_terms(int64_t n, const double *__restrict__ abc, const double *__restrict__ w, double *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double c = abc[3 * i + 2];
    out[5 * i] = abc[3 * i]; out[5 * i + 1] = abc[3 * i + 1];
    out[5 * i + 2] = c * w[3 * i]; out[5 * i + 3] = c * w[3 * i + 1]; out[5 * i + 4] = c * w[3 * i + 2];
}

// stage 0 + stage 1 (the pointwise local solves, tb_sarcomere.hip) ahead of the element kernels
template <class FE> static int condensed_prepass(tb_form *f, const double *d_u, bool need_tangent)
{
    tb_mesh *m = f->mesh;
    tb_device *dev = m->dev;
    const int64_t npts = m->n_cells * FE::NQ;
    if (!f->d_Q || !f->d_Qknown) { set_error("condensed internal variable: no internal state set (tb_hyperelastic_set_internal_state)"); return TB_ERR_BAD_ARG; }
    if (!(f->cond_dt > 0.0)) { set_error("condensed internal variable: the time step must be positive (got %g)", f->cond_dt); return TB_ERR_BAD_ARG; }
    const bool rate = f->d_u_prev != nullptr;
    // layout: λ | Ca | records (2 or 5 per point; rate-free: at 2·npts) | [rate: velocity | w (3) | (a,b,c) (3)] | status
    const size_t ndbl = rate ? 14 : 4;
    if (!f->d_qp_buf) {
        hipError_t e = hipMalloc((void **)&f->d_qp_buf, sizeof(double) * ndbl * (size_t)npts + sizeof(int32_t) * (size_t)npts);
        if (e != hipSuccess) { set_error("quadrature-point buffers: %s", hipGetErrorString(e)); return TB_ERR_NOMEM; }
        f->d_qp_act = f->d_qp_buf + 2 * npts;
        TB_HIP(hipMemsetAsync(f->d_qp_buf, 0, sizeof(double) * ndbl * (size_t)npts + sizeof(int32_t) * (size_t)npts, dev->stream)); // points outside a subdomain stay zero / TB_LOCAL_SUCCESS
    }
    double *lam = f->d_qp_buf, *ca = f->d_qp_buf + npts;
    double *vel = rate ? f->d_qp_buf + 7 * npts : nullptr, *w = rate ? f->d_qp_buf + 8 * npts : nullptr, *abc = rate ? f->d_qp_buf + 11 * npts : nullptr;
    int32_t *status = (int32_t *)(f->d_qp_buf + (rate ? 14 : 4) * npts);
    const MechMesh mm{m->d_xyz, m->d_conn, m->d_cell_dofs, f->d_field, nullptr, nullptr, 2, nullptr};
    const int64_t ncell_work = f->has_cellset ? f->n_set : m->n_cells;
    if (ncell_work == 0) { f->cond_n_failed = 0; return TB_OK; }
    hipLaunchKernelGGL(k_fiber_stretch<FE>, dim3((unsigned)ncell_work), dim3(64), 0, dev->stream, mm, make_params(f), f->d_act_field, f->act_tension, d_u, lam, ca,
                       f->d_u_prev, 1.0 / f->cond_dt, vel, w, dev->d_status, f->has_cellset ? f->d_cellset : nullptr);
    TB_HIP(hipGetLastError());
    int64_t nfail = 0;
    int rc = launch_sarcomere_implicit(dev, f->cond_params, f->d_Q, f->d_Qknown, npts, lam, vel, ca, 0.0, 0.0, 0.0, f->cond_dt, f->cond_tol, f->cond_max_iters,
                                       f->cond_tmax, nullptr, nullptr, rate ? abc : f->d_qp_act, rate ? 3 : 2, status, &nfail, need_tangent,
                                       f->has_cellset ? f->d_cellset : nullptr, FE::NQ, f->n_set);
    if (rc) return rc;
    if (rate) {
        hipLaunchKernelGGL(k_pack_rate_terms, dim3((unsigned)((npts + 255) / 256)), dim3(256), 0, dev->stream, npts, abc, w, f->d_qp_act);
        TB_HIP(hipGetLastError());
    }
    f->cond_n_failed = nfail;
    return rc;
}

int launch_hyperelastic(tb_form *f, tb_pattern *p, int strategy, const double *d_u, double *d_nz, double *d_r)
{
    tb_mesh *m = f->mesh;
    int rc = reset_status(m->dev);
    if (rc) return rc;
    const bool q2 = m->field_kind == TB_HEX27;
    if (!q2 && !(m->field_kind == TB_HEX8 && f->qorder == 2)) { set_error("hyperelastic: Q1 field needs quadrature order 2"); return TB_ERR_UNSUPPORTED; }
    if (q2 && f->qorder != 3) { set_error("hyperelastic: Q2 field needs quadrature order 3"); return TB_ERR_UNSUPPORTED; }
    // Q2 tangents: sum-factorised contraction (default); TB_MECH_CONTRACT=mfma selects the matrix-core build, =vector the plain FMA sweep over
    // the points (comparison builds; TB_MECH_MFMA=0 is the older spelling of "vector")
    const char *ce = tune_env("TB_MECH_CONTRACT");
    const bool old_vec = tune_env("TB_MECH_MFMA") && atoi(tune_env("TB_MECH_MFMA")) == 0;
    const int ct2 = (ce && !strcmp(ce, "mfma")) ? 1 : ((ce && !strcmp(ce, "vector")) || old_vec) ? 0 : 2;
    const bool ad = !form_is_fast_path(f);
    if (f->cond_model) {
        rc = q2 ? condensed_prepass<Q2Vec>(f, d_u, d_nz != nullptr) : condensed_prepass<Q1Vec>(f, d_u, d_nz != nullptr);
        if (rc) return rc;
    }
    const bool nonsym = f->cond_model && f->d_u_prev; // rate-coupled internal variable: the symmetric-tile matrix-core kernel does not apply
    const int ct = !q2 ? 0 : (ct2 == 1 && nonsym) ? 0 : ct2;
#define TB_RUN(FEV, K, R, C) (ad ? run<FEV, K, R, C, true>(f, p, strategy, d_u, d_nz, d_r) : run<FEV, K, R, C, false>(f, p, strategy, d_u, d_nz, d_r))
    if (d_nz && d_r) rc = !q2 ? TB_RUN(Q1Vec, true, true, 0) : ct == 2 ? TB_RUN(Q2Vec, true, true, 2) : ct == 1 ? TB_RUN(Q2Vec, true, true, 1) : TB_RUN(Q2Vec, true, true, 0);
    else if (d_nz) rc = !q2 ? TB_RUN(Q1Vec, true, false, 0) : ct == 2 ? TB_RUN(Q2Vec, true, false, 2) : ct == 1 ? TB_RUN(Q2Vec, true, false, 1) : TB_RUN(Q2Vec, true, false, 0);
    else rc = q2 ? TB_RUN(Q2Vec, false, true, 0) : TB_RUN(Q1Vec, false, true, 0);
#undef TB_RUN
    if (rc) return rc;
    return check_status(m->dev);
}


// ------------------------------------------------------------------------------------------------ weak boundary conditions
// Facet integrals of src/modeling/core/weak_boundary_conditions.jl (RobinBC :102-198, NormalSpringBC :200-300, pressure
// follower load :419-515).  One 64-thread workgroup per (cell, local facet): per facet Gauss point the shape values,
// mapped gradients and the few tensors the integrand needs are parked in LDS, then every thread sums its (i, j) pairs over
// the points and adds the result to the CSR entry / residual entry with one atomic each (surface terms are O(n²) work, the
// volume term O(n³): this kernel is not on the critical path).  FacetValues conventions are Ferrite's (tbhip.h).
__host__ __device__ constexpr int facet_fix(int lf) { constexpr int v[6] = {2, 1, 0, 1, 0, 2}; return v[lf]; }
__host__ __device__ constexpr double facet_val(int lf) { constexpr double v[6] = {-1, -1, 1, 1, -1, 1}; return v[lf]; }
__host__ __device__ constexpr int facet_s(int lf) { constexpr int v[6] = {1, 0, 1, 2, 2, 0}; return v[lf]; }
__host__ __device__ constexpr int facet_t(int lf) { constexpr int v[6] = {0, 2, 2, 0, 1, 1}; return v[lf]; }

template <int NB>
__device__ inline void shape_at(int a, const double (&xi)[3], double &N, double (&dN)[3])
{
    if (NB == 8) {
        const double f[3] = {1.0 + hex_sgn(a, 0) * xi[0], 1.0 + hex_sgn(a, 1) * xi[1], 1.0 + hex_sgn(a, 2) * xi[2]};
        N = 0.125 * f[0] * f[1] * f[2];
        dN[0] = 0.125 * hex_sgn(a, 0) * f[1] * f[2];
        dN[1] = 0.125 * f[0] * hex_sgn(a, 1) * f[2];
        dN[2] = 0.125 * f[0] * f[1] * hex_sgn(a, 2);
    } else {
        double v[3], d[3];
        for (int k = 0; k < 3; ++k) { v[k] = quad1d(hex27_tix(a, k), xi[k]); d[k] = dquad1d(hex27_tix(a, k), xi[k]); }
        N = v[0] * v[1] * v[2];
        dN[0] = d[0] * v[1] * v[2];
        dN[1] = v[0] * d[1] * v[2];
        dN[2] = v[0] * v[1] * d[2];
    }
}

template <int NB>
__global__ void __launch_bounds__(64)
k_facets(MechMesh m, const int32_t *__restrict__ facets, int bc, double param, int fq, const double *__restrict__ pfield, const double *__restrict__ u, double *__restrict__ nz,
         double *__restrict__ r, const int64_t *__restrict__ rowptr, const uint16_t *__restrict__ blockpos, Status *st)
{
    constexpr int ND = 3 * NB, MAXQ = 9;
    const int tid = threadIdx.x;
    const int64_t cell = facets[2 * blockIdx.x];
    const int lf = facets[2 * blockIdx.x + 1];
    __shared__ double s_ue[ND], s_x[24], s_N[MAXQ][NB], s_G[MAXQ][NB][3];
    __shared__ double s_q[MAXQ][32]; // per point: dΓ, n₀[3], grad[3] (Robin / spring) or J·cofF·n₀ [3], H[9] / invF[9], J, invFᵀ… (see below)
    __shared__ int32_t s_dof[ND];
    for (int i = tid; i < ND; i += 64) { const int32_t d = m.cell_dofs[cell * ND + i]; s_dof[i] = d; s_ue[i] = u[d]; }
    for (int i = tid; i < 24; i += 64) s_x[i] = m.xyz[3 * (int64_t)m.conn[cell * 8 + i / 3] + i % 3];
    __syncthreads();
    const int nq = fq * fq;
    const double gx[3][3] = {{0.0, 0.0, 0.0}, {-0.5773502691896258, 0.5773502691896258, 0.0}, {-0.7745966692414834, 0.0, 0.7745966692414834}};
    const double gw[3][3] = {{2.0, 0.0, 0.0}, {1.0, 1.0, 0.0}, {0.5555555555555556, 0.8888888888888888, 0.5555555555555556}};
    // geometry + shape values per point: thread (q, a)
    for (int idx = tid; idx < nq * NB; idx += 64) {
        const int q = idx / NB, a = idx % NB;
        double xi[3];
        xi[facet_fix(lf)] = facet_val(lf);
        xi[facet_s(lf)] = gx[fq - 1][q % fq];
        xi[facet_t(lf)] = gx[fq - 1][q / fq];
        double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        for (int v = 0; v < 8; ++v) {
            double Mv, dM[3];
            shape_at<8>(v, xi, Mv, dM);
            for (int i = 0; i < 3; ++i) for (int k = 0; k < 3; ++k) J[i][k] += s_x[3 * v + i] * dM[k];
        }
        const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1], c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2], c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
        const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02, id = 1.0 / det;
        const double Ji[3][3] = {{c00 * id, (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id, (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * id},
                                 {c01 * id, (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id, (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id},
                                 {c02 * id, (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * id, (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * id}};
        double Na, dNa[3];
        shape_at<NB>(a, xi, Na, dNa);
        s_N[q][a] = Na;
        for (int k = 0; k < 3; ++k) s_G[q][a][k] = dNa[0] * Ji[0][k] + dNa[1] * Ji[1][k] + dNa[2] * Ji[2][k];
        if (a == 0) {
            if (!(det > 0.0)) { st->neg_detj = 1; st->cell = cell; }
            const int cs = facet_s(lf), ct = facet_t(lf);
            const double av[3] = {J[0][cs], J[1][cs], J[2][cs]}, bv[3] = {J[0][ct], J[1][ct], J[2][ct]};
            const double nw[3] = {av[1] * bv[2] - av[2] * bv[1], av[2] * bv[0] - av[0] * bv[2], av[0] * bv[1] - av[1] * bv[0]};
            const double len = sqrt(nw[0] * nw[0] + nw[1] * nw[1] + nw[2] * nw[2]);
            s_q[q][0] = len * gw[fq - 1][q % fq] * gw[fq - 1][q / fq];
            for (int k = 0; k < 3; ++k) s_q[q][1 + k] = nw[k] / len;
        }
    }
    __syncthreads();
    // field values per point (one thread per point): u_q, F, and the tensors of the integrand
    //   slots: [4..6] g = residual vector density (δuᵢ·g), [7..15] H (Robin / spring: Hessian; pressure: invF), [16] J, [17..19] cofF·n₀,
    //          [20..22] invFᵀ-weighted normal  fin[d] = Σ_r invF[r][d] n₀[r]
    if (tid < nq) {
        const int q = tid;
        double uq[3] = {0, 0, 0}, F[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
        for (int a = 0; a < NB; ++a)
            for (int c = 0; c < 3; ++c) {
                uq[c] += s_N[q][a] * s_ue[3 * a + c];
                for (int k = 0; k < 3; ++k) F[c][k] += s_ue[3 * a + c] * s_G[q][a][k];
            }
        const double *n0 = &s_q[q][1];
        double *o = s_q[q];
        if (bc == TB_BC_ROBIN) {
            for (int c = 0; c < 3; ++c) o[4 + c] = 2.0 * param * uq[c];
            for (int e = 0; e < 9; ++e) o[7 + e] = (e % 4 == 0) ? 2.0 * param : 0.0;
        } else if (bc == TB_BC_NORMAL_SPRING) {
            const double un = uq[0] * n0[0] + uq[1] * n0[1] + uq[2] * n0[2];
            for (int c = 0; c < 3; ++c) o[4 + c] = param * un * n0[c];
            for (int c = 0; c < 3; ++c) for (int d = 0; d < 3; ++d) o[7 + 3 * c + d] = param * n0[c] * n0[d];
        } else {
            double pq = param;
            if (bc == TB_BC_PRESSURE_FIELD && pfield) { // evaluate_coefficient(pc, cell, qp, t): nodal data × M_a at the facet point
                double xi[3];
                xi[facet_fix(lf)] = facet_val(lf);
                xi[facet_s(lf)] = gx[fq - 1][q % fq];
                xi[facet_t(lf)] = gx[fq - 1][q / fq];
                double v = 0.0;
                for (int a = 0; a < 8; ++a) { double Ma, dMa[3]; shape_at<8>(a, xi, Ma, dMa); v += Ma * pfield[cell * 8 + a]; }
                pq = param * v;
            }
            const double c00 = F[1][1] * F[2][2] - F[1][2] * F[2][1], c01 = F[1][2] * F[2][0] - F[1][0] * F[2][2], c02 = F[1][0] * F[2][1] - F[1][1] * F[2][0];
            const double Jf = F[0][0] * c00 + F[0][1] * c01 + F[0][2] * c02, id = 1.0 / Jf;
            const double Fi[9] = {c00 * id, (F[0][2] * F[2][1] - F[0][1] * F[2][2]) * id, (F[0][1] * F[1][2] - F[0][2] * F[1][1]) * id,
                                  c01 * id, (F[0][0] * F[2][2] - F[0][2] * F[2][0]) * id, (F[0][2] * F[1][0] - F[0][0] * F[1][2]) * id,
                                  c02 * id, (F[0][1] * F[2][0] - F[0][0] * F[2][1]) * id, (F[0][0] * F[1][1] - F[0][1] * F[1][0]) * id};
            for (int e = 0; e < 9; ++e) o[7 + e] = Fi[e];
            o[16] = bc == TB_BC_BENDING_SPRING ? Jf : pq * Jf;      // pressure: p·J folded here
            double vv[3];
            for (int c = 0; c < 3; ++c) {
                const double cn = Fi[0 + c] * n0[0] + Fi[3 + c] * n0[1] + Fi[6 + c] * n0[2]; // v = F⁻ᵀ n₀
                vv[c] = cn;
                o[17 + c] = cn;
                o[4 + c] = pq * Jf * cn;
            }
            if (bc == TB_BC_BENDING_SPRING) { // w = v − N, z = F⁻¹ w, B = F⁻¹ F⁻ᵀ (slots 20..22, 23..31)
                const double w[3] = {vv[0] - n0[0], vv[1] - n0[1], vv[2] - n0[2]};
                for (int j = 0; j < 3; ++j) o[20 + j] = Fi[3 * j + 0] * w[0] + Fi[3 * j + 1] * w[1] + Fi[3 * j + 2] * w[2];
                for (int j = 0; j < 3; ++j)
                    for (int l = 0; l < 3; ++l) o[23 + 3 * j + l] = Fi[3 * j + 0] * Fi[3 * l + 0] + Fi[3 * j + 1] * Fi[3 * l + 1] + Fi[3 * j + 2] * Fi[3 * l + 2];
            }
        }
    }
    __syncthreads();
    // residual: rₑ[i] += δuᵢ·g dΓ   (bending spring: ∇δuᵢ ⊡ P dΓ with P = −kᵇ v ⊗ z)
    if (r)
        for (int i = tid; i < ND; i += 64) {
            const int a = i / 3, c = i % 3;
            double v = 0.0;
            if (bc == TB_BC_BENDING_SPRING) {
                for (int q = 0; q < nq; ++q) {
                    const double *o = s_q[q], *g = s_G[q][a];
                    v -= param * o[17 + c] * (g[0] * o[20] + g[1] * o[21] + g[2] * o[22]) * o[0];
                }
            } else {
                for (int q = 0; q < nq; ++q) v += s_N[q][a] * s_q[q][4 + c] * s_q[q][0];
            }
            if (v != 0.0) unsafeAtomicAdd(r + s_dof[i], v);
        }
    // tangent
    if (nz)
        for (int ij = tid; ij < ND * ND; ij += 64) {
            const int i = ij / ND, j = ij % ND, a = i / 3, c = i % 3, b = j / 3, d = j % 3;
            double v = 0.0;
            if (bc == TB_BC_ROBIN || bc == TB_BC_NORMAL_SPRING) {
                for (int q = 0; q < nq; ++q) v += s_N[q][a] * s_q[q][7 + 3 * c + d] * s_N[q][b] * s_q[q][0];
            } else if (bc == TB_BC_BENDING_SPRING) {
                // 𝔸[c][k][d][l] = kᵇ [ v_d F⁻¹_lc z_k + v_c (F⁻¹_kd z_l + v_d B_kl) ],  Kₑ += ∇N_a[k] 𝔸 ∇N_b[l] dΓ
                for (int q = 0; q < nq; ++q) {
                    const double *o = s_q[q], *Fi = o + 7, *ga = s_G[q][a], *gb = s_G[q][b], *z = o + 20, *B = o + 23;
                    const double gaz = ga[0] * z[0] + ga[1] * z[1] + ga[2] * z[2], gbz = gb[0] * z[0] + gb[1] * z[1] + gb[2] * z[2];
                    const double gaFd = ga[0] * Fi[0 + d] + ga[1] * Fi[3 + d] + ga[2] * Fi[6 + d];
                    const double gbFc = gb[0] * Fi[0 + c] + gb[1] * Fi[3 + c] + gb[2] * Fi[6 + c];
                    double gaBgb = 0.0;
                    for (int k = 0; k < 3; ++k) gaBgb += ga[k] * (B[3 * k] * gb[0] + B[3 * k + 1] * gb[1] + B[3 * k + 2] * gb[2]);
                    v += param * (o[17 + d] * gbFc * gaz + o[17 + c] * (gaFd * gbz + o[17 + d] * gaBgb)) * o[0];
                }
            } else {
                for (int q = 0; q < nq; ++q) {
                    const double *o = s_q[q], *Fi = o + 7, *g = s_G[q][b];
                    double gF[3];
                    for (int k = 0; k < 3; ++k) gF[k] = g[0] * Fi[0 + k] + g[1] * Fi[3 + k] + g[2] * Fi[6 + k];
                    // δJ·cofF·n₀ + J·δcofF·n₀ for δF = e_d ⊗ ∇N_b
                    v += o[16] * (gF[d] * o[17 + c] - gF[c] * o[17 + d]) * s_N[q][a] * o[0];
                }
            }
            if (v != 0.0) unsafeAtomicAdd(nz + rowptr[s_dof[3 * a] + c] + blockpos[cell * (NB * NB) + a * NB + b] + d, v);
        }
}

int launch_facets(tb_form *f, tb_pattern *p, const double *d_u, double *d_nz, double *d_r)
{
    tb_mesh *m = f->mesh;
    tb_device *dev = m->dev;
    int rc = reset_status(dev);
    if (rc) return rc;
    if (d_nz) { rc = ensure_blockpos(p); if (rc) return rc; }
    const MechMesh mm{m->d_xyz, m->d_conn, m->d_cell_dofs, nullptr, nullptr};
    const int64_t *rowptr = p ? p->d_rowptr : nullptr;
    const uint16_t *bp = p ? p->d_blockpos : nullptr;
    if (m->field_kind == TB_HEX27)
        hipLaunchKernelGGL((k_facets<27>), dim3((unsigned)f->n_facets), dim3(64), 0, dev->stream, mm, f->d_facets, f->bc_kind, f->bc_param, f->facet_q, f->d_field,
                           d_u, d_nz, d_r, rowptr, bp, dev->d_status);
    else
        hipLaunchKernelGGL((k_facets<8>), dim3((unsigned)f->n_facets), dim3(64), 0, dev->stream, mm, f->d_facets, f->bc_kind, f->bc_param, f->facet_q, f->d_field,
                           d_u, d_nz, d_r, rowptr, bp, dev->d_status);
    TB_HIP(hipGetLastError());
    return check_status(dev);
}

} // namespace tb
