// tb_mech_split.hip — the triquadratic hyperelastic tangent of the element strategy as a point kernel + a contraction kernel.
#include <hip/hip_runtime.h>

#include "tb_internal.h"
#include "tb_energy.hpp"
#include "tb_material.hpp"
#include "tb_math.hpp"
#include "tb_mech_common.hpp"
#include "tb_mech_split.hpp"

namespace tb {
using namespace tbk;

// ------------------------------------------------------------------------------------------------------------------------------------
// Round 5: the triquadratic tangent of the element strategy in TWO kernels (TB_MECH_SPLIT=0 selects the fused kernel above).
// Phase A of k_hyperelastic — unknowns, Jacobians, F, the material, the pull-backs — is work per QUADRATURE POINT, yet inside a
// one-cell workgroup it runs on 27 or 243 lanes of 256 behind eight barriers and two dependent trips to memory: 11.4 of the
// 19.4 µs of a cell (profiles/r03_v2/mechanics_phase_stamps_80.txt).  Here it is a kernel of its own with one LANE per point
// (k_mech_points: nine cells = 243 points per workgroup, everything in registers, no barrier after the load phase), which leaves
// per cell a record of the pulled-back tangent Â_q (symmetric: 45 of its 81 entries) and stress P̂_q at the 27 points —
// 54 × 27 doubles = 11.7 KB, [entry][point] — and the contraction kernel (k_mech_contract, one cell per workgroup) starts from ONE
// coalesced read of that record: stages 1–3 of the sum factorisation and the stores of Kₑ, nothing else.  Same sums as
// elements.jl:211-223; 𝔸's major symmetry (materials.jl:1025-1040: a Hessian) is used, so Kₑ is exactly symmetric.
#ifndef TB_MECH_CONTRACT_WAVES
#define TB_MECH_CONTRACT_WAVES 4
#endif
constexpr int MP_CELLS = 9;          // cells per workgroup of k_mech_points
__host__ __device__ constexpr int sympair(int m, int n) { return m * 9 - m * (m - 1) / 2 + (n - m); } // m ≤ n < 9 → 0..44

#ifndef TB_MECH_POINTS_WAVES
#define TB_MECH_POINTS_WAVES 1 // one wave per SIMD, no scratch: 2.6 ms at 80³ against 3.6 ms at two waves with 74 spilled registers
#endif
// AD: any energy of tb_energy.hpp by hyper-dual evaluation (the reference's Tensors.hessian route, materials.jl:1025-1040) with directional seeds in Ĥ —
// 45 evaluations per point in a loop, each yielding one record entry directly; no LDS, the registers of a lone wave (the fused kernel's AD instances
// spill ≈ 1 000 registers at their 168-register budget).
#ifndef TB_MECH_AD_WAVES
#define TB_MECH_AD_WAVES 1
#endif
#ifndef TB_MECH_AD_GROUPS
#define TB_MECH_AD_GROUPS 1 // (measured: 1 → 8.2, 5 → 10.5, 9 → 13.0 ms for Guccione at 40³: every group repeats the kinematics)  // the 45 pair evaluations of a point are dealt to this many workgroups (blockIdx.y): more waves in flight for a kernel of long dependent chains
#endif
template <bool AD, int EN = -1, int HA = -1>
__global__ void __launch_bounds__(256, AD ? TB_MECH_AD_WAVES : TB_MECH_POINTS_WAVES)
k_mech_points(MechMesh m, HOParams mat, EnergyParams en, const double *__restrict__ u, int64_t n_cells /*of this launch, from m.cell0*/, double *__restrict__ out, Status *st)
{
    const int tid = threadIdx.x;
    const int64_t rel0 = (int64_t)blockIdx.x * MP_CELLS;
    __shared__ double s_ue[MP_CELLS * 81], s_x[MP_CELLS * 24];
    // load_element_unknowns! (elements.jl:125-132) of the nine cells: dof ids and vertex coordinates are contiguous over consecutive cells
    for (int i = tid; i < MP_CELLS * 81; i += 256)
        if (rel0 + i / 81 < n_cells) s_ue[i] = u[m.cell_dofs[(m.cell0 + rel0) * 81 + i]];
    for (int i = tid; i < MP_CELLS * 24; i += 256)
        if (rel0 + i / 24 < n_cells) s_x[i] = m.cell_xyz[(m.cell0 + rel0) * 24 + i];
    __syncthreads();
    const int cl = tid / 27, q = tid - 27 * cl;
    if (tid >= MP_CELLS * 27 || rel0 + cl >= n_cells) return;
    const int64_t cell = m.cell0 + rel0 + cl;
    const int qd[3] = {q % 3, (q / 3) % 3, q / 9};
    // J, J⁻¹, dΩ (PR883.jl:253-263,367-387): ∂M/∂ξ from the vertex signs and the 1-D rule
    double mm_[3], pp_[3], wq = 1.0, PH[3][3], DP[3][3]; // PH[dim][i] = φᵢ(ξ_dim), DP = φᵢ′
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const double xi = qd[d] == 0 ? G3::x(0) : qd[d] == 1 ? G3::x(1) : G3::x(2);
        mm_[d] = 1.0 - xi; pp_[d] = 1.0 + xi;
        wq *= qd[d] == 1 ? G3::w(1) : G3::w(0);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            PH[d][i] = qd[d] == 0 ? quad1d(i, G3::x(0)) : qd[d] == 1 ? quad1d(i, G3::x(1)) : quad1d(i, G3::x(2));
            DP[d][i] = qd[d] == 0 ? dquad1d(i, G3::x(0)) : qd[d] == 1 ? dquad1d(i, G3::x(1)) : dquad1d(i, G3::x(2));
        }
    }
    const double *X = s_x + cl * 24;
    double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, Mv[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        const double f0 = hex_sgn(a, 0) > 0 ? pp_[0] : mm_[0], f1 = hex_sgn(a, 1) > 0 ? pp_[1] : mm_[1], f2 = hex_sgn(a, 2) > 0 ? pp_[2] : mm_[2];
        const double dm[3] = {0.125 * hex_sgn(a, 0) * f1 * f2, 0.125 * hex_sgn(a, 1) * f0 * f2, 0.125 * hex_sgn(a, 2) * f0 * f1};
        Mv[a] = 0.125 * f0 * f1 * f2;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double xa = X[3 * a + i];
#pragma unroll
            for (int k = 0; k < 3; ++k) J[i][k] += xa * dm[k];
        }
    }
    const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1], c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2], c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
    const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02, id = 1.0 / det;
    double ji[9];
    ji[0] = c00 * id; ji[1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id; ji[2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * id;
    ji[3] = c01 * id; ji[4] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id; ji[5] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id;
    ji[6] = c02 * id; ji[7] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * id; ji[8] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * id;
    const double dO = det * wq;
    if (!(dO > 0.0)) { st->neg_detj = 1; st->cell = cell; }
    // Ĥ[c][s] = Σₐ uₐ[c] ∂̂ₛNₐ(ξ_q), one direction after the other (the point is fixed per lane, so the 1-D factors are lane constants)
    const double *ue = s_ue + cl * 81;
    double F[3][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double h[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int a2 = 0; a2 < 3; ++a2) {
            double g0 = 0.0, g1 = 0.0, g2 = 0.0; // Σ over a₁ (and a₀) with the derivative in direction 0 / 1 / none
#pragma unroll
            for (int a1 = 0; a1 < 3; ++a1) {
                double tp = 0.0, td = 0.0;
#pragma unroll
                for (int a0 = 0; a0 < 3; ++a0) {
                    const double uv = ue[3 * make_hex27_nodes().v[a0 + 3 * a1 + 9 * a2] + c];
                    tp += PH[0][a0] * uv; td += DP[0][a0] * uv;
                }
                g0 += PH[1][a1] * td; g1 += DP[1][a1] * tp; g2 += PH[1][a1] * tp;
            }
            h[0] += PH[2][a2] * g0; h[1] += PH[2][a2] * g1; h[2] += DP[2][a2] * g2;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) F[c][k] = (c == k ? 1.0 : 0.0) + h[0] * ji[k] + h[1] * ji[3 + k] + h[2] * ji[6 + k];
    }
    HOParams mq = mat;
    if (m.act_field) { // Ta(x_q) = Tmax·Σₐ Mₐ(ξ_q)·state[cell][a] (coefficients.jl:85-99, contraction.jl:166-175)
        double ca = 0.0;
#pragma unroll
        for (int a = 0; a < 8; ++a) ca += Mv[a] * m.act_field[cell * 8 + a];
        mq.Ta = mat.Ta * ca;
    }
    if (m.qp_act) { mq.Ta = m.qp_act[m.qp_stride * (cell * 27 + q)]; mq.Tb = m.qp_act[m.qp_stride * (cell * 27 + q) + 1]; }
    if (m.fsn_field) { // interpolate the nodal frame, normalise, Gram–Schmidt (microstructure.jl:176-187)
        double f[3] = {0, 0, 0}, s[3] = {0, 0, 0}, n[3] = {0, 0, 0};
        const double *fc = m.fsn_field + cell * 72;
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int d = 0; d < 3; ++d) { f[d] += Mv[a] * fc[9 * a + d]; s[d] += Mv[a] * fc[9 * a + 3 + d]; n[d] += Mv[a] * fc[9 * a + 6 + d]; }
        ho_orthonormal_frame(f, s, n);
#pragma unroll
        for (int d = 0; d < 3; ++d) { mq.f[d] = f[d]; mq.s[d] = s[d]; mq.n[d] = n[d]; }
    }
    double Ff[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) Ff[e] = F[e / 3][e % 3];
    double *o = out + (rel0 + cl) * QP_REC + q;
    if constexpr (AD) {
        const double Ta = mq.Ta; // uniform tension × nodal activation, or the condensed (a) of this point
        const double f0[3] = {mq.f[0], mq.f[1], mq.f[2]}, s0[3] = {mq.s[0], mq.s[1], mq.s[2]}, n0[3] = {mq.n[0], mq.n[1], mq.n[2]};
        // condensed internal variable: + b (∂λ/∂Ĥ)_m (∂λ/∂Ĥ)_n with ∂λ/∂Ĥ[c][s] = g_c (f₀·J⁻¹[s][·]) / ‖g‖, g = F f₀
        const double g[3] = {F[0][0] * f0[0] + F[0][1] * f0[1] + F[0][2] * f0[2], F[1][0] * f0[0] + F[1][1] * f0[1] + F[1][2] * f0[2],
                             F[2][0] * f0[0] + F[2][1] * f0[1] + F[2][2] * f0[2]};
        const double fj[3] = {ji[0] * f0[0] + ji[1] * f0[1] + ji[2] * f0[2], ji[3] * f0[0] + ji[4] * f0[1] + ji[5] * f0[2], ji[6] * f0[0] + ji[7] * f0[1] + ji[8] * f0[2]};
        const double tb_over_gg = mq.Tb != 0.0 ? mq.Tb / (g[0] * g[0] + g[1] * g[1] + g[2] * g[2]) : 0.0;
        auto sel3 = [](double a, double b, double c, int k) { return k == 0 ? a : k == 1 ? b : c; };
        constexpr int PPG = (QP_SYM + TB_MECH_AD_GROUPS - 1) / TB_MECH_AD_GROUPS;
        const int pr_begin = (int)blockIdx.y * PPG, pr_end = pr_begin + PPG < QP_SYM ? pr_begin + PPG : QP_SYM;
#pragma unroll 1
        for (int pr = pr_begin; pr < pr_end; ++pr) {
            int mm, nn;
            pair_components(pr, mm, nn);
            const int c = mm / 3, s_ = mm - 3 * c, d = nn / 3, u_ = nn - 3 * d;
            double da[9], db[9];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    da[3 * i + j] = i == c ? sel3(ji[j], ji[3 + j], ji[6 + j], s_) : 0.0;
                    db[3 * i + j] = i == d ? sel3(ji[j], ji[3 + j], ji[6 + j], u_) : 0.0;
                }
            const HD r = energy_pair_dir<EN, HA>(en, Ff, da, db, f0, s0, n0, Ta);
            double ab = r.ab;
            if (tb_over_gg != 0.0) ab += tb_over_gg * (sel3(g[0], g[1], g[2], c) * sel3(fj[0], fj[1], fj[2], s_)) * (sel3(g[0], g[1], g[2], d) * sel3(fj[0], fj[1], fj[2], u_));
            o[pr * 27] = ab * dO;
            if (mm == nn) o[(QP_SYM + mm) * 27] = r.a * dO;
        }
        return;
    }
    double C[HOC_SIZE];
    ho_common<false>(mq, F, C);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double row[3][9], Pc[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) ho_row<true>(mq, C, Ff, c, k, dO, Pc[k], row[k]);
        // P̂[c][s] = Σ_k J⁻¹[s][k] P[c][k]·dΩ
#pragma unroll
        for (int s_ = 0; s_ < 3; ++s_) o[(QP_SYM + 3 * c + s_) * 27] = ji[3 * s_] * Pc[0] + ji[3 * s_ + 1] * Pc[1] + ji[3 * s_ + 2] * Pc[2];
        // Â[c][s][d][u] = Σ_kl J⁻¹[s][k] 𝔸[c][k][d][l] J⁻¹[u][l]·dΩ, the pairs (c,s) ≤ (d,u) only
#pragma unroll
        for (int d = c; d < 3; ++d) {
            double t9[3][3];
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int u_ = 0; u_ < 3; ++u_) t9[k][u_] = row[k][3 * d] * ji[3 * u_] + row[k][3 * d + 1] * ji[3 * u_ + 1] + row[k][3 * d + 2] * ji[3 * u_ + 2];
#pragma unroll
            for (int s_ = 0; s_ < 3; ++s_)
#pragma unroll
                for (int u_ = 0; u_ < 3; ++u_)
                    if (3 * c + s_ <= 3 * d + u_)
                        o[sympair(3 * c + s_, 3 * d + u_) * 27] = ji[3 * s_] * t9[0][u_] + ji[3 * s_ + 1] * t9[1][u_] + ji[3 * s_ + 2] * t9[2][u_];
        }
    }
}

// stages 1–3 of the sum-factorised contraction (comment in k_hyperelastic) from the records of k_mech_points; one cell per workgroup, Kₑ / rₑ stored
// for the gather of the element strategy.  37 KB of LDS: four workgroups per CU where the fused kernel has three.
// SYM: Kₑ leaves symmetric-packed by rank (block (i, j), i ≤ j, at symblk(i, j)·9 as [c][d] seen from the node of rank i; the lower blocks are not stored —
// the gather reads them transposed): half the bytes of the 81 × 81 form, the largest stream of a linearisation.
template <bool NEED_R, bool SYM>
__global__ void __launch_bounds__(256, TB_MECH_CONTRACT_WAVES)
k_mech_contract(const double *__restrict__ qp /*records of this launch*/, int64_t cell0, double *__restrict__ ke, double *__restrict__ re, const uint8_t *__restrict__ rank27)
{
    constexpr int ND = 81, NQ = 27;
    const int tid = threadIdx.x;
    const int64_t cell = cell0 + blockIdx.x;
#ifndef TB_MECH_CONTRACT_PAD
#define TB_MECH_CONTRACT_PAD 0
#endif
    __shared__ double s_A[NQ][81], s_Z1[2187 + TB_MECH_CONTRACT_PAD], s_P[NQ][9];
    { // the record [entry][point] → 𝔸̂ of every point with both triangles (s_A[q][9m + n] = s_A[q][9n + m]) and P̂: a lane keeps its point, six entries each
        const double *src = qp + (int64_t)blockIdx.x * QP_REC;
        const int q = tid % 27, e0 = tid / 27; // e0 ≤ 9 (lanes 243..255: e0 = 9, their entries e0 + 9i ≥ 54 for i = 5 only)
        double v[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) { const int e = e0 + 9 * i; v[i] = e < QP_ENT ? src[e * 27 + q] : 0.0; }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int e = e0 + 9 * i;
            if (e < QP_SYM) {
                int mrow = 0;
#pragma unroll
                for (int t = 1; t < 9; ++t) mrow += e >= sympair(t, t);
                const int ncol = mrow + (e - (mrow * 9 - mrow * (mrow - 1) / 2));
                s_A[q][9 * mrow + ncol] = v[i];
                s_A[q][9 * ncol + mrow] = v[i];
            } else if (e < QP_ENT) {
                s_P[q][e - QP_SYM] = v[i];
            }
        }
    }
    __syncthreads();
    auto PH = [](int i, int q) constexpr { return quad1d(i, G3::x(q)); };
    auto DP = [](int i, int q) constexpr { return dquad1d(i, G3::x(q)); };
    double racc = 0.0;
    if constexpr (NEED_R) { // rₑ[(a,c)] = Σ_q Σ_s ∂̂ₛNₐ(ξ_q) P̂_q[c][s]
        if (tid < ND) {
            const int a = tid / 3, c = tid - 3 * a;
            double fa[3][3], da[3][3]; // [direction][point]
            const int ta = tix27(a);
#pragma unroll
            for (int dir = 0; dir < 3; ++dir) {
                const int i = dir == 0 ? ta % 3 : dir == 1 ? (ta / 3) % 3 : ta / 9;
#pragma unroll
                for (int qq = 0; qq < 3; ++qq) {
                    fa[dir][qq] = i == 0 ? PH(0, qq) : i == 1 ? PH(1, qq) : PH(2, qq);
                    da[dir][qq] = i == 0 ? DP(0, qq) : i == 1 ? DP(1, qq) : DP(2, qq);
                }
            }
#pragma unroll
            for (int q2 = 0; q2 < 3; ++q2)
#pragma unroll
                for (int q1 = 0; q1 < 3; ++q1) {
                    const double m12 = fa[1][q1] * fa[2][q2], d1 = da[1][q1] * fa[2][q2], d2 = fa[1][q1] * da[2][q2];
#pragma unroll
                    for (int q0 = 0; q0 < 3; ++q0) {
                        const double *pp = s_P[q0 + 3 * q1 + 9 * q2] + 3 * c;
                        racc += da[0][q0] * m12 * pp[0] + fa[0][q0] * (d1 * pp[1] + d2 * pp[2]);
                    }
                }
        }
    }
    double *Z1 = s_Z1;
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
    const int t27 = (tid < 243 ? tid : 0) % 27; // 9·d + 3·b₀ + b₁: the lane's column inside a b₂ block of the tensor-order row
    double *kc = ke + (int64_t)cell * (SYM ? KE_SYM : ND * ND);
    int ra[3] = {0, 0, 0}, rb[3] = {0, 0, 0}; // SYM: ranks of the lane's row nodes (a₀, a₁, ·) and column nodes (b₀, b₁, ·)
    if constexpr (SYM) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            ra[k] = rank27[cell * 32 + g_hex27_node[ta0 + 3 * ta1 + 9 * k]];
            rb[k] = rank27[cell * 32 + g_hex27_node[tb0 + 3 * tb1 + 9 * k]];
        }
    }
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
                    if constexpr (SYM) { if (ra[a2] <= rb[b2]) kc[symblk(ra[a2], rb[b2]) * 9 + 3 * c + td] = v; } // (diagonal blocks: all nine entries, from their own lanes)
                    else kc[(3 * (ta0 + 3 * ta1 + 9 * a2) + c) * ND + 27 * b2 + t27] = v; // entry ((a, c), (b, d)) in the tensor-order layout (tb_mech_common.hpp): 27 consecutive lanes, 27 consecutive doubles
                }
        }
        lds_barrier(); // Z1 is rewritten by the next component's stage 1
    }
    if constexpr (NEED_R) {
        if (tid < ND) re[cell * ND + tid] = racc;
    }
}


int launch_mech_points(tb_device *dev, const MechMesh &mm, const HOParams &hp, const EnergyParams *ep, const double *d_u, int64_t n, double *d_qp)
{
    if (!n) return TB_OK;
    const dim3 grid((unsigned)((n + MP_CELLS - 1) / MP_CELLS));
    if (ep) {
        const dim3 g2(grid.x, TB_MECH_AD_GROUPS);
#define TB_MP(E) hipLaunchKernelGGL((k_mech_points<true, E>), g2, dim3(256), 0, dev->stream, mm, hp, *ep, d_u, n, d_qp, dev->d_status)
#define TB_MPH(E, A) hipLaunchKernelGGL((k_mech_points<true, E, A>), g2, dim3(256), 0, dev->stream, mm, hp, *ep, d_u, n, d_qp, dev->d_status)
        // Generalized / Extended Hill models: the (passive energy, active spring) pairings of the reference's tests and tutorials at compile time
        const bool hill_ct = ep->hill != HILL_NONE && !ep->prestressed;
        if (hill_ct && ep->energy == EN_HOLZAPFEL_OGDEN && ep->act_energy == EN_LINEAR_SPRING) TB_MPH(EN_HOLZAPFEL_OGDEN, EN_LINEAR_SPRING);
        else if (hill_ct && ep->energy == EN_HOLZAPFEL_OGDEN && ep->act_energy == ACT_SIMPLE_ACTIVE_SPRING) TB_MPH(EN_HOLZAPFEL_OGDEN, ACT_SIMPLE_ACTIVE_SPRING);
        else if (hill_ct && ep->energy == EN_LIN_YIN_PASSIVE && ep->act_energy == EN_LIN_YIN_ACTIVE) TB_MPH(EN_LIN_YIN_PASSIVE, EN_LIN_YIN_ACTIVE);
        else if (hill_ct && ep->energy == EN_GUCCIONE_1991 && ep->act_energy == EN_LINEAR_SPRING) TB_MPH(EN_GUCCIONE_1991, EN_LINEAR_SPRING);
        else if (ep->hill != HILL_NONE || ep->prestressed) TB_MP(-1); // other active energies, prestress: the run-time form
        else switch (ep->energy) {
            case EN_NULL: TB_MP(EN_NULL); break;
            case EN_BIO_NEOHOOKEAN: TB_MP(EN_BIO_NEOHOOKEAN); break;
            case EN_TI_NEOHOOKEAN: TB_MP(EN_TI_NEOHOOKEAN); break;
            case EN_LIN_YIN_PASSIVE: TB_MP(EN_LIN_YIN_PASSIVE); break;
            case EN_LIN_YIN_ACTIVE: TB_MP(EN_LIN_YIN_ACTIVE); break;
            case EN_HUMPHREY_STRUMPF_YIN: TB_MP(EN_HUMPHREY_STRUMPF_YIN); break;
            case EN_LINEAR_SPRING: TB_MP(EN_LINEAR_SPRING); break;
            case EN_GUCCIONE_1991: TB_MP(EN_GUCCIONE_1991); break;
            default: TB_MP(EN_HOLZAPFEL_OGDEN); break; // (with another penalty than the fast path's)
        }
#undef TB_MP
#undef TB_MPH
    }
    else hipLaunchKernelGGL(k_mech_points<false>, grid, dim3(256), 0, dev->stream, mm, hp, EnergyParams{}, d_u, n, d_qp, dev->d_status);
    TB_HIP(hipGetLastError());
    return TB_OK;
}

int launch_mech_contract(tb_device *dev, const double *d_qp, int64_t cell0, int64_t n, double *d_ke, double *d_re, const uint8_t *d_rank27)
{
    if (!n) return TB_OK;
#define TB_MC(R, S) hipLaunchKernelGGL((k_mech_contract<R, S>), dim3((unsigned)n), dim3(256), 0, dev->stream, d_qp, cell0, d_ke, d_re, d_rank27)
#ifdef TB_ABLATION // symmetric-packed storage: measured slower (tb_mechanics.hip), instantiated in the profiling build only
    if (d_rank27) { if (d_re) TB_MC(true, true); else TB_MC(false, true); }
    else
#endif
    { if (d_re) TB_MC(true, false); else TB_MC(false, false); }
#undef TB_MC
    TB_HIP(hipGetLastError());
    return TB_OK;
}

} // namespace tb
