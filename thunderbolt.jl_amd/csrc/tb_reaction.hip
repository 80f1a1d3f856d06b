// tb_reaction.hip — pointwise ionic-ODE reaction step (one thread per point, states in registers).
//
// Restates src/solver/time/partitioned_solver.jl:80-99 (forward Euler) and :196-234 (adaptive
// sub-stepper) with the cell right-hand sides of src/modeling/cells/fhn.jl:21-34,
// src/modeling/cells/aliev-panfilov.jl:17-31 and src/modeling/cells/pcg2019.jl:52-133.
// Replaces the single in-tree CUDA kernel ext/CuThunderboltExt.jl:103-124.
//
// HBM-bound by design: every state is read once and written once per step (16 B per DoF-update,
// 24 B when `du` is materialised for the RTC controller); parameters travel as a kernel argument
// (SGPRs / scalar cache), never through global memory.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "tb_internal.h"
#include "tb_math.hpp"

namespace tb {

struct CellParams {
    double p[48];
};

template <int MODEL> struct CellModel;

template <> struct CellModel<TB_CELL_FHN> {
    static constexpr int NS = 2, PHI = 0;
    static constexpr bool HAS_GATES = false;
    __device__ __forceinline__ static void rhs(const CellParams &P, const double (&u)[NS], double, double (&du)[NS])
    {
        const double a = P.p[0], b = P.p[1], c = P.p[2], d = P.p[3], e = P.p[4], f = P.p[5];
        const double phi = u[0], s = u[1];
        du[0] = f * (phi * (1 - phi) * (phi - a) - s);
        du[1] = e * (b * phi - c * s - d);
    }
};

template <> struct CellModel<TB_CELL_ALIEV_PANFILOV> {
    static constexpr int NS = 2, PHI = 1;
    static constexpr bool HAS_GATES = false; // state order (s, φₘ)
    __device__ __forceinline__ static void rhs(const CellParams &P, const double (&u)[NS], double, double (&du)[NS])
    {
        const double ct = P.p[0], k = P.p[1], a = P.p[2], e0 = P.p[3], mu1 = P.p[4], mu2 = P.p[5];
        const double phi = u[1], s = u[0];
        const double eps = e0 + s * mu1 / (phi + mu2);
        du[1] = ct * (k * phi * (phi - 1.0) * (phi - a) - phi * s);
        du[0] = ct * eps * (-s - k * phi * (phi - a - 1.0));
    }
};

template <> struct CellModel<TB_CELL_PCG2019> {
    static constexpr int NS = 7, PHI = 0;
    static constexpr bool HAS_GATES = true; // every state but φₘ relaxes as (g∞ − g)/τ_g (pcg2019.jl:96-118)
    // parameter slots follow the struct field order of pcg2019.jl:4-48
    enum { gNa, Em, km, taum, Eh, kh, dh, tauh0, gK1, Ez, kz, gto, Er, kr, Es, ks, taus, gCaL, Ed, kd, Ef, kf, tauf,
           gKr, Exr, kxr, tauxr, Ey, ky, gKs, Exs, kxs, tauxs, ENa, EK, ECa };
    // 1/(1 + exp(±(φ − E)/k)).  Divisions by parameters are multiplications by their reciprocals (loop invariants of the point loop; an IEEE FP64
    // division is ≈ 15 instructions, and there were 18 of them per evaluation besides the 12 exponentials), 1/(1 + e) is a refined hardware reciprocal
    __device__ __forceinline__ static double sigmoid(double phi, double E, double ik, double sign)
    {
        return rcp_b(1.0 + exp_b(sign * (phi - E) * ik));
    }
    __device__ __forceinline__ static void rhs(const CellParams &P, const double (&u)[NS], double t, double (&du)[NS])
    {
        double rate[NS];
        rhs_rates(P, u, t, du, rate);
    }
    // rate[k] = 1/τ of gate k (0 for φₘ): the Rush–Larsen step advances gates with the exact solution of their linear ODE for frozen φₘ
    __device__ __forceinline__ static void rhs_rates(const CellParams &P, const double (&u)[NS], double, double (&du)[NS], double (&rate)[NS])
    {
        const double *p = P.p;
        const double phi = u[0], h = u[1], m = u[2], f = u[3], s = u[4], xs = u[5], xr = u[6];
        const double rinf = sigmoid(phi, p[Er], 1.0 / p[kr], -1.0);
        const double dinf = sigmoid(phi, p[Ed], 1.0 / p[kd], -1.0);
        const double zinf = sigmoid(phi, p[Ez], 1.0 / p[kz], 1.0);
        const double yinf = sigmoid(phi, p[Ey], 1.0 / p[ky], 1.0);
        const double I_Na = p[gNa] * m * m * m * h * h * (phi - p[ENa]);
        const double I_K1 = p[gK1] * zinf * (phi - p[EK]);
        const double I_to = p[gto] * rinf * s * (phi - p[EK]);
        const double I_CaL = p[gCaL] * dinf * f * (phi - p[ECa]);
        const double I_Kr = p[gKr] * xr * yinf * (phi - p[EK]);
        const double I_Ks = p[gKs] * xs * (phi - p[EK]);
        du[0] = -(I_Na + I_K1 + I_to + I_CaL + I_Kr + I_Ks); // C_m = 1, pcg2019.jl:55
        const double xh = (phi - p[Eh]) * (1.0 / p[kh]);
        const double eh = exp_b(xh);
        const double hinf = rcp_b(1.0 + eh);
        // τ_h = 2 τ_h0 exp(δ_h x)/(1 + e^x)  →  1/τ_h = (1 + e^x)·exp(−δ_h x)/(2 τ_h0)
        rate[0] = 0.0; rate[1] = (1.0 + eh) * exp_b(-p[dh] * xh) * (0.5 / p[tauh0]); rate[2] = 1.0 / p[taum]; rate[3] = 1.0 / p[tauf]; rate[4] = 1.0 / p[taus];
        rate[5] = 1.0 / p[tauxs]; rate[6] = 1.0 / p[tauxr];
        du[1] = (hinf - h) * rate[1];
        du[2] = (sigmoid(phi, p[Em], 1.0 / p[km], -1.0) - m) * rate[2];
        du[3] = (sigmoid(phi, p[Ef], 1.0 / p[kf], 1.0) - f) * rate[3];
        du[4] = (sigmoid(phi, p[Es], 1.0 / p[ks], 1.0) - s) * rate[4];
        du[5] = (sigmoid(phi, p[Exs], 1.0 / p[kxs], -1.0) - xs) * rate[5];
        du[6] = (sigmoid(phi, p[Exr], 1.0 / p[kxr], -1.0) - xr) * rate[6];
    }
};

// HeterogeneousFHNModel of the reference's how-to (docs/src/literate-howto/custom-ep-cell-model.jl:8-56): FitzHugh–Nagumo whose recovery rate e
// depends on the point's coordinate x — there a Julia closure e(x,t); closures cannot cross a C ABI, so the bound form is the
// affine field e(x) = e₀ + g·x (parameters a, b, c, d, e₀, gₓ, g_y, g_z).  The one built-in model that reads `x` (partitioned_solver.jl:88-92).
template <> struct CellModel<TB_CELL_FHN_HETEROGENEOUS> {
    static constexpr int NS = 2, PHI = 0;
    static constexpr bool HAS_GATES = false, USES_X = true;
    __device__ __forceinline__ static void rhs_x(const CellParams &P, const double (&u)[NS], const float (&x)[3], double, double (&du)[NS])
    {
        const double a = P.p[0], b = P.p[1], c = P.p[2], d = P.p[3];
        const double e = P.p[4] + P.p[5] * (double)x[0] + P.p[6] * (double)x[1] + P.p[7] * (double)x[2];
        const double phi = u[0], s = u[1];
        du[0] = phi * (1.0 - phi) * (phi - a) - s;
        du[1] = e * (b * phi - c * s - d);
    }
};


// ten Tusscher & Panfilov 2006 human ventricular model, epicardial parameters (Am J Physiol 291:H1088) — BASELINE.json's
// config 3 names it, the reference does not have it: an EXTENSION whose parity is pinned only against this
// repository's own oracle.  Plain ODE right-hand side for cell_rhs!-style explicit stepping; the m gate (τ ≈ 1 µs·10³ at
// rest) needs Δt ≲ 2·10⁻³ ms under forward Euler — use the sub-stepper.
template <> struct CellModel<TB_CELL_TT06> {
    static constexpr int NS = 19, PHI = 0;
    // The ≈50 exponentials of the published right-hand side are affine in V with only a handful of distinct slopes:
    // exp_b((V + c)/k) = exp_b(V/k)·exp_b(c/k), exp_b(−x) = 1/exp_b(x).  One exp per slope plus constant factors (folded at compile
    // time) and a reciprocal replaces 3–5 library calls each; the kernel is exp-bound (FP64 exp ≈ 40 instructions), so this
    // is where its time goes.  Divisions by literal constants are written as multiplications by the (compile-time) reciprocal
    // for the same reason (an IEEE FP64 division is ≈15 instructions).  Rounding differs from the literal form by a few ulp
    // (parity tolerance 1e-12).
    __device__ __forceinline__ static double sgm(double e) { return rcp_b(1.0 + e); } // 1/(1+e), e = exp_b(x) given
    static constexpr bool HAS_GATES = true;
    // rate[k] = 1/τ_k for the Hodgkin–Huxley-type gates (dy/dt = (y∞ − y)/τ, y∞ and τ functions of V / Ca only), 0 otherwise:
    // what a Rush–Larsen step needs besides du (see k_reaction_rl)
    __device__ __forceinline__ static void rhs(const CellParams &P, const double (&u)[NS], double t, double (&du)[NS])
    {
        double rate[NS];
        rhs_rates(P, u, t, du, rate);
    }
    __device__ __forceinline__ static void rhs_rates(const CellParams &P, const double (&u)[NS], double, double (&du)[NS], double (&rate)[NS])
    {
#pragma unroll
        for (int k = 0; k < NS; ++k) rate[k] = 0.0;
        const double *p = P.p;
        const double GNa = p[0], GK1 = p[1], GKr = p[2], GKs = p[3], Gto = p[4], GCaL = p[5], GbNa = p[6], GbCa = p[7], GpCa = p[8],
                     GpK = p[9], knak = p[10], knaca = p[11], Ko = p[12], Cao = p[13], Nao = p[14], Vc = p[15], Vsr = p[16], Vss = p[17],
                     Bufc = p[18], Kbufc = p[19], Bufsr = p[20], Kbufsr = p[21], Bufss = p[22], Kbufss = p[23], Vmaxup = p[24], Kup = p[25],
                     Vrel = p[26], k1p = p[27], k2p = p[28], k3 = p[29], k4 = p[30], EC = p[31], maxsr = p[32], minsr = p[33],
                     Vleak = p[34], Vxfer = p[35], Cm = p[36], pKNa = p[37], KmK = p[38], KmNa = p[39], KmNai = p[40], KmCa = p[41],
                     ksat = p[42], nn = p[43], KpCa = p[44], R = p[45], T = p[46], F = p[47];
        const double V = u[0], Cai = u[1], CaSR = u[2], CaSS = u[3], Nai = u[4], Ki = u[5], m = u[6], h = u[7], j = u[8], xr1 = u[9],
                     xr2 = u[10], xs = u[11], r = u[12], s = u[13], d = u[14], f = u[15], f2 = u[16], fCass = u[17], RR = u[18];
        const double RTONF = R * T / F, FoRT = F / (R * T);
        // reversal potentials
        // log(a/b) = log a − log b: the numerators are parameters (their logarithms are loop invariants), the state-dependent ones take the bounded form
        const double Ek = RTONF * (log(Ko) - log_b(Ki)), Ena = RTONF * (log(Nao) - log_b(Nai));
        const double Eks = RTONF * (log(Ko + pKNa * Nao) - log_b(Ki + pKNa * Nai)), Eca = 0.5 * RTONF * (log(Cao) - log_b(Cai));
        // shared exponentials of V (p = exp_b(V/k), q = 1/p)
        const double p5 = exp_b(V * (1.0 / 5.0)), q5 = rcp_b(p5);
        const double p10 = exp_b(V * (1.0 / 10.0)), q10 = rcp_b(p10);
        const double p20 = exp_b(V * (1.0 / 20.0)), q20 = rcp_b(p20);
        const double p7 = exp_b(V * (1.0 / 7.0)), q7 = rcp_b(p7);
        const double q6 = exp_b(-V * (1.0 / 6.0));
        const double w = exp_b(V * FoRT), iw = rcp_b(w); // exp_b(±V F/RT)
        // currents
        const double vk = V - Ek;
        const double Ak1 = 0.1 * sgm(exp_b(0.06 * (vk - 200.0)));
        const double eh = exp_b(-0.5 * vk);
        const double Bk1 = (3.0 * exp_b(0.0002 * (vk + 100.0)) + exp_b(0.1 * (vk - 10.0))) * sgm(eh);
        const double INa = GNa * m * m * m * h * j * (V - Ena);
        const double e2 = w * w * exp_b(-30.0 * FoRT); // exp_b(2 (V − 15) F/RT)
        const double ICaL = GCaL * d * f * f2 * fCass * 4.0 * (V - 15.0) * (F * FoRT) * (0.25 * e2 * CaSS - Cao) * rcp_b(e2 - 1.0);
        const double Ito = Gto * r * s * vk;
        const double IKr = GKr * sqrt(Ko * (1.0 / 5.4)) * xr1 * xr2 * vk;
        const double IKs = GKs * xs * xs * (V - Eks);
        const double IK1 = GK1 * (Ak1 * rcp_b(Ak1 + Bk1)) * vk;
        const double en = exp_b(nn * V * FoRT), en1 = en * iw; // exp_b((γ − 1) V F/RT)
        const double INaCa = knaca * (1.0 / (KmNai * KmNai * KmNai + Nao * Nao * Nao)) * (1.0 / (KmCa + Cao)) * rcp_b(1.0 + ksat * en1) *
                             (en * Nai * Nai * Nai * Cao - en1 * Nao * Nao * Nao * Cai * 2.5);
        const double INaK = knak * (Ko / (Ko + KmK)) * (Nai * rcp_b(Nai + KmNa)) * rcp_b(1.0 + 0.1245 * exp_b(-0.1 * V * FoRT) + 0.0353 * iw);
        const double IpCa = GpCa * Cai * rcp_b(KpCa + Cai);
        const double IpK = GpK * sgm(exp_b((25.0 - V) * (1.0 / 5.98))) * vk;
        const double IbNa = GbNa * (V - Ena), IbCa = GbCa * (V - Eca);
        du[0] = -(IKr + IKs + IK1 + Ito + INa + IbNa + ICaL + IbCa + INaK + INaCa + IpCa + IpK);
        // calcium handling
        const double ecs = EC * rcp_b(CaSR);
        const double kCaSR = maxsr - (maxsr - minsr) * rcp_b(1.0 + ecs * ecs);
        const double k1 = k1p * rcp_b(kCaSR), k2 = k2p * kCaSR;
        du[18] = k4 * (1.0 - RR) - k2 * CaSS * RR;
        const double O = k1 * CaSS * CaSS * RR * rcp_b(k3 + k1 * CaSS * CaSS);
        const double Irel = Vrel * O * (CaSR - CaSS), Ileak = Vleak * (CaSR - Cai);
        // 1/(1 + a/x²) = x²/(x² + a): one reciprocal each
        const double Iup = Vmaxup * (Cai * Cai) * rcp_b(Cai * Cai + Kup * Kup), Ixfer = Vxfer * (CaSS - Cai);
        const double sc = (Cai + Kbufc) * (Cai + Kbufc), ssr = (CaSR + Kbufsr) * (CaSR + Kbufsr), sss = (CaSS + Kbufss) * (CaSS + Kbufss);
        const double bc = sc * rcp_b(sc + Bufc * Kbufc), bsr = ssr * rcp_b(ssr + Bufsr * Kbufsr), bss = sss * rcp_b(sss + Bufss * Kbufss);
        du[1] = bc * ((Ileak - Iup) * Vsr / Vc + Ixfer - (IbCa + IpCa - 2.0 * INaCa) * Cm / (2.0 * Vc * F));
        du[2] = bsr * (Iup - Irel - Ileak);
        du[3] = bss * (-ICaL * Cm / (2.0 * Vss * F) + Irel * Vsr / Vss - Ixfer * Vc / Vss);
        du[4] = -(INa + IbNa + 3.0 * INaK + 3.0 * INaCa) * Cm / (Vc * F);
        du[5] = -(IK1 + Ito + IKr + IKs - 2.0 * INaK + IpK) * Cm / (Vc * F);
        // gates: dy/dt = (y∞ − y)/τ.  Constants below are exp_b(c/k) of the published arguments (V + c)/k, folded by the compiler.
        const double AM = sgm(q5 * 6.14421235332821e-06 /* exp_b(-60.0 * (1.0 / 5.0)) */);
        const double BM = 0.1 * sgm(p5 * 1096.6331584284585 /* exp_b(35.0 * (1.0 / 5.0)) */) + 0.1 * sgm(exp_b((V - 50.0) * (1.0 / 200.0)));
        const double mr = sgm(exp_b((-56.86 - V) * (1.0 / 9.03)));
        rate[6] = rcp_b(AM * BM);
        du[6] = (mr * mr - m) * rate[6];
        const double hr = sgm(exp_b((V + 71.55) * (1.0 / 7.43)));
        const double hinf = hr * hr;
        double ABH, ABJ; // αh + βh, αj + βj
        if (V >= -40.0) {
            ABH = (0.77 / 0.13) * sgm(exp_b(-(V + 10.66) * (1.0 / 11.1)));
            ABJ = 0.6 * exp_b(0.057 * V) * sgm(q10 * 0.04076220397836621 /* exp_b(-3.2) */);
        } else {
            ABH = 0.057 * exp_b(-(V + 80.0) * (1.0 / 6.8)) + 2.7 * exp_b(0.079 * V) + 3.1e5 * exp_b(0.3485 * V);
            ABJ = ((-2.5428e4) * exp_b(0.2444 * V) - 6.948e-6 * exp_b(-0.04391 * V)) * (V + 37.78) * sgm(exp_b(0.311 * (V + 79.23))) +
                  0.02424 * exp_b(-0.01052 * V) * sgm(exp_b(-0.1378 * (V + 40.14)));
        }
        rate[7] = ABH;
        du[7] = (hinf - h) * ABH;
        rate[8] = ABJ;
        du[8] = (hinf - j) * ABJ;
        rate[9] = rcp_b((450.0 * sgm(q10 * 0.011108996538242306 /* exp_b(-4.5) */)) * (6.0 * sgm(exp_b((V + 30.0) * (1.0 / 11.5)))));
        du[9] = (sgm(q7 * 0.02437284407327961 /* exp_b(-26.0 * (1.0 / 7.0)) */) - xr1) * rate[9];
        rate[10] = rcp_b((3.0 * sgm(q20 * 0.049787068367863944 /* exp_b(-3.0) */)) * (1.12 * sgm(p20 * 0.049787068367863944 /* exp_b(-3.0) */)));
        du[10] = (sgm(exp_b((V + 88.0) * (1.0 / 24.0))) - xr2) * rate[10];
        rate[11] = rcp_b((1400.0 / sqrt(1.0 + q6 * 2.3009758908928246 /* exp_b(5.0 * (1.0 / 6.0)) */)) * sgm(exp_b((V - 35.0) * (1.0 / 15.0))) + 80.0);
        du[11] = (sgm(exp_b((-5.0 - V) * (1.0 / 14.0))) - xs) * rate[11];
        rate[12] = rcp_b(9.5 * exp_b(-(V + 40.0) * (V + 40.0) * (1.0 / 1800.0)) + 0.8);
        du[12] = (sgm(q6 * 28.031624894526125 /* exp_b(20.0 * (1.0 / 6.0)) */) - r) * rate[12];
        rate[13] = rcp_b(85.0 * exp_b(-(V + 45.0) * (V + 45.0) * (1.0 / 320.0)) + 5.0 * sgm(p5 * 0.01831563888873418 /* exp_b(-4.0) */) + 3.0);
        du[13] = (sgm(p5 * 54.598150033144236 /* exp_b(4.0) */) - s) * rate[13];
        rate[14] = rcp_b((1.4 * sgm(exp_b((-35.0 - V) * (1.0 / 13.0))) + 0.25) * (1.4 * sgm(p5 * 2.718281828459045 /* exp_b(1.0) */)) + sgm(q20 * 12.182493960703473 /* exp_b(2.5) */));
        du[14] = (sgm(exp_b((-8.0 - V) * (1.0 / 7.5))) - d) * rate[14];
        rate[15] = rcp_b((1102.5 * exp_b(-(V + 27.0) * (V + 27.0) * (1.0 / 225.0)) + 200.0 * sgm(q10 * 3.6692966676192444 /* exp_b(1.3) */) + 180.0 * sgm(p10 * 20.085536923187668 /* exp_b(3.0) */) + 20.0));
        du[15] = (sgm(p7 * 17.411708063327644 /* exp_b(20.0 * (1.0 / 7.0)) */) - f) * rate[15];
        rate[16] = rcp_b((600.0 * exp_b(-(V + 25.0) * (V + 25.0) * (1.0 / 170.0)) + 31.0 * sgm(q10 * 12.182493960703473 /* exp_b(2.5) */) + 16.0 * sgm(p10 * 20.085536923187668 /* exp_b(3.0) */)));
        du[16] = (0.67 * sgm(p7 * 148.4131591025766 /* exp_b(5.0) */) + 0.33 - f2) * rate[16];
        const double cq = 1.0 + (CaSS * (1.0 / 0.05)) * (CaSS * (1.0 / 0.05));
        const double icq = rcp_b(cq);
        rate[17] = rcp_b(80.0 * icq + 2.0);
        du[17] = (0.6 * icq + 0.4 - fCass) * rate[17];
    }
};


// O'Hara, Virág, Varró & Rudy 2011 (PLoS Comput Biol 7(5): e1002061), human ventricular model, 41 states in the order of its published code:
// v, nai, nass, ki, kss, cai, cass, cansr, cajsr, m, hf, hs, j, hsp, jp, mL, hL, hLp, a, iF, iS, ap, iFp, iSp, d, ff, fs, fcaf, fcas, jca, nca, ffp,
// fcafp, xrf, xrs, xs1, xs2, xk1, Jrelnp, Jrelp, CaMKt.  EXTENSION (SURVEY §8 f4 names the model, the reference has only the reaction_rhs! /
// state_rhs! hooks, src/modeling/cells/fhn.jl:36-60); the oracle's restatement is pinned on the paper's endocardial numbers
// (tests/test_ord_known_answer.py) and this routine on the oracle (1e-12 per step).  Parameters: thirteen conductance scalings (1 = published),
// nao, cao, ko, cell type.  1/(1 + e^x) are refined hardware reciprocals, x^1.4 = exp(1.4 log x), the integer powers are products.
template <> struct CellModel<TB_CELL_ORD11> {
    static constexpr int NS = 41, PHI = 0;
    static constexpr bool HAS_GATES = true;
    __device__ __forceinline__ static double sg(double x) { return rcp_b(1.0 + exp_b(x)); } // 1/(1 + e^x)
    __device__ __forceinline__ static void rhs(const CellParams &P, const double (&u)[NS], double t, double (&du)[NS])
    {
        double rate[NS];
        rhs_rates(P, u, t, du, rate);
    }
    // The right-hand side in two parts, so that a kernel can retire the 13 non-gate states before it touches the dynamics of the 28 gates (below):
    // currents(): everything that reads gate VALUES — the currents, fluxes and the rates of V, the concentrations, nca, Jrel and CaMK (du[0…8], du[30],
    // du[38…40]); gates(): steady states and rate constants of the gates as functions of V alone, handed one by one to `emit(k, y∞, 1/τ)`.
    __device__ __forceinline__ static void currents(const CellParams &P, const double (&u)[NS], double (&du)[NS])
    {
        const double *p = P.p;
        const double nao = p[13], cao = p[14], ko = p[15];
        const int celltype = (int)p[16];
        constexpr double R = 8314.0, T = 310.0, F = 96485.0, RTF = R * T / F, FRT = F / (R * T);
        constexpr double L = 0.01, rad = 0.0011, vcell = 1000 * 3.14 * rad * rad * L, Ageo = 2 * 3.14 * rad * rad + 2 * 3.14 * rad * L, Acap = 2 * Ageo;
        constexpr double vmyo = 0.68 * vcell, vnsr = 0.0552 * vcell, vjsr = 0.0048 * vcell, vss = 0.02 * vcell;
        const double v = u[0], nai = u[1], nass = u[2], ki = u[3], kss = u[4], cai = u[5], cass = u[6], cansr = u[7], cajsr = u[8], m = u[9], hf = u[10],
                     hs = u[11], j = u[12], hsp = u[13], jp = u[14], mL = u[15], hL = u[16], hLp = u[17], a = u[18], iF = u[19], iS = u[20],
                     ap = u[21], iFp = u[22], iSp = u[23], d = u[24], ff = u[25], fs = u[26], fcaf = u[27], fcas = u[28], jca = u[29], nca = u[30],
                     ffp = u[31], fcafp = u[32], xrf = u[33], xrs = u[34], xs1 = u[35], xs2 = u[36], xk1 = u[37], Jrelnp = u[38], Jrelp = u[39],
                     CaMKt = u[40];
        // CaMK
        constexpr double KmCaMK = 0.15, aCaMK = 0.05, bCaMK = 0.00068, CaMKo = 0.05, KmCaM = 0.0015;
        const double CaMKb = CaMKo * (1.0 - CaMKt) * cass * rcp_b(cass + KmCaM), CaMKa = CaMKb + CaMKt;
        du[40] = aCaMK * CaMKb * (CaMKb + CaMKt) - bCaMK * CaMKt;
        const double fpa = CaMKa * rcp_b(CaMKa + KmCaMK), fnp = 1.0 - fpa;
        // reversal potentials
        const double ENa = RTF * (log(nao) - log(nai)), EK = RTF * (log(ko) - log(ki));
        const double EKs = RTF * (log(ko + 0.01833 * nao) - log(ki + 0.01833 * nai));
        const double vg = fabs(v) < 1e-7 ? 1e-7 : v; // the constant-field fluxes divide by e^{zVF/RT} − 1: V = 0 exactly is moved by 10⁻⁷ mV (as in the oracle)
        const double vfrt = vg * FRT, vffrt = vfrt * F;
        // INa
        constexpr double Ahf = 0.99, Ahs = 1.0 - Ahf;
        const double h = Ahf * hf + Ahs * hs;
        const double hp = Ahf * hf + Ahs * hsp;
        const double INa = 75.0 * p[0] * (v - ENa) * m * m * m * (fnp * h * j + fpa * hp * jp);
        // INaL
        const double INaL = 0.0075 * p[1] * (celltype == 1 ? 0.6 : 1.0) * (v - ENa) * mL * (fnp * hL + fpa * hLp);
        // Ito
        const double AiF = sg((v - 213.6) * (1.0 / 151.2)), AiS = 1.0 - AiF;
        const double i_ = AiF * iF + AiS * iS;
        const double ip = AiF * iFp + AiS * iSp;
        const double Ito = 0.02 * p[2] * (celltype == 0 ? 1.0 : 4.0) * (v - EK) * (fnp * a * i_ + fpa * ap * ip);
        // ICaL, ICaNa, ICaK
        constexpr double Aff = 0.6, Afs = 1.0 - Aff;
        const double f = Aff * ff + Afs * fs;
        const double Afcaf = 0.3 + 0.6 * sg((v - 10.0) * (1.0 / 10.0)), Afcas = 1.0 - Afcaf;
        const double fca = Afcaf * fcaf + Afcas * fcas;
        const double fp = Aff * ffp + Afs * fs;
        const double fcap = Afcaf * fcafp + Afcas * fcas;
        constexpr double Kmn = 0.002, k2n = 1000.0;
        const double km2n = jca;
        const double q1 = 1.0 + Kmn * rcp_b(cass), q2 = q1 * q1;
        const double anca = rcp_b(k2n * rcp_b(km2n) + q2 * q2);
        du[30] = anca * k2n - nca * km2n;
        const double e1 = exp_b(vfrt), e2 = e1 * e1;
        const double ie1 = rcp_b(e1 - 1.0), ie2 = rcp_b(e2 - 1.0);
        const double PhiCaL = 4.0 * vffrt * (cass * e2 - 0.341 * cao) * ie2;
        const double PhiCaNa = vffrt * (0.75 * nass * e1 - 0.75 * nao) * ie1;
        const double PhiCaK = vffrt * (0.75 * kss * e1 - 0.75 * ko) * ie1;
        const double PCa = 0.0001 * p[3] * (celltype == 1 ? 1.2 : celltype == 2 ? 2.5 : 1.0), PCap = 1.1 * PCa;
        const double gnp = d * (f * (1.0 - nca) + jca * fca * nca), gp = d * (fp * (1.0 - nca) + jca * fcap * nca);
        const double gsum = fnp * PCa * gnp + fpa * PCap * gp; // the permeability ratios PCaNa / PCa, PCaK / PCa are the same for both populations
        const double ICaL = gsum * PhiCaL, ICaNa = 0.00125 * gsum * PhiCaNa, ICaK = 3.574e-4 * gsum * PhiCaK;
        // IKr
        const double Axrf = sg((v + 54.81) * (1.0 / 38.21)), Axrs = 1.0 - Axrf;
        const double xr = Axrf * xrf + Axrs * xrs;
        const double rkr = sg((v + 55.0) * (1.0 / 75.0)) * sg((v - 10.0) * (1.0 / 30.0));
        const double IKr = 0.046 * p[4] * (celltype == 1 ? 1.3 : celltype == 2 ? 0.8 : 1.0) * sqrt(ko * (1.0 / 5.4)) * xr * rkr * (v - EK);
        // IKs
        const double KsCa = 1.0 + 0.6 * rcp_b(1.0 + exp_b(1.4 * (log(3.8e-5) - log(cai))));
        const double IKs = 0.0034 * p[5] * (celltype == 1 ? 1.4 : 1.0) * KsCa * xs1 * xs2 * (v - EKs);
        // IK1
        const double rk1 = sg((v + 105.8 - 2.6 * ko) * (1.0 / 9.493));
        const double IK1 = 0.1908 * p[6] * (celltype == 1 ? 1.2 : celltype == 2 ? 1.3 : 1.0) * sqrt(ko) * rk1 * xk1 * (v - EK);
        // INaCa: the same six-state cycle in the myoplasm (80 %) and in the subspace (20 %)
        constexpr double kna1 = 15.0, kna2 = 5.0, kna3 = 88.12, kasymm = 12.5, wna = 6.0e4, wca = 6.0e4, wnaca = 5.0e3, kcaon = 1.5e6, kcaoff = 5.0e3,
                         qna = 0.5224, qca = 0.1670, KmCaAct = 150.0e-6;
        const double hca = exp_b(qca * vfrt), hna = exp_b(qna * vfrt), ihna = rcp_b(hna);
        const double Gncx = 0.0008 * p[7] * (celltype == 1 ? 1.1 : celltype == 2 ? 1.4 : 1.0);
        const double h7 = 1.0 + nao * (1.0 / kna3) * (1.0 + ihna), h8 = nao * ihna * rcp_b(kna3 * h7), h9 = rcp_b(h7);
        const double h10 = kasymm + 1.0 + nao * (1.0 / kna1) * (1.0 + nao * (1.0 / kna2)), h11 = nao * nao / (h10 * kna1 * kna2), h12 = 1.0 / h10;
        const double k1 = h12 * cao * kcaon, k3p = h9 * wca, k3pp = h8 * wnaca, k3 = k3p + k3pp, k8 = h8 * h11 * wna;
        double INaCa_c[2];
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            const double na = w == 0 ? nai : nass, ca = w == 0 ? cai : cass;
            const double h1 = 1.0 + na * (1.0 / kna3) * (1.0 + hna), h3 = rcp_b(h1), h2 = na * hna * (1.0 / kna3) * h3;
            const double h4 = 1.0 + na * (1.0 / kna1) * (1.0 + na * (1.0 / kna2)), h6 = rcp_b(h4), h5 = na * na * (1.0 / (kna1 * kna2)) * h6;
            const double k2 = kcaoff, k4p = h3 * wca * rcp_b(hca), k4pp = h2 * wnaca, k4 = k4p + k4pp, k5 = kcaoff, k6 = h6 * ca * kcaon, k7 = h5 * h2 * wna;
            const double x1 = k2 * k4 * (k7 + k6) + k5 * k7 * (k2 + k3), x2 = k1 * k7 * (k4 + k5) + k4 * k6 * (k1 + k8);
            const double x3 = k1 * k3 * (k7 + k6) + k8 * k6 * (k2 + k3), x4 = k2 * k8 * (k4 + k5) + k3 * k5 * (k1 + k8);
            const double ixs = rcp_b(x1 + x2 + x3 + x4), E1 = x1 * ixs, E2 = x2 * ixs, E3 = x3 * ixs, E4 = x4 * ixs;
            const double allo = ca * ca * rcp_b(ca * ca + KmCaAct * KmCaAct);
            const double JncxNa = 3.0 * (E4 * k7 - E1 * k8) + E3 * k4pp - E2 * k3pp, JncxCa = E2 * k2 - E1 * k1;
            INaCa_c[w] = (w == 0 ? 0.8 : 0.2) * Gncx * allo * (JncxNa + 2.0 * JncxCa);
        }
        const double INaCa_i = INaCa_c[0], INaCa_ss = INaCa_c[1];
        // INaK
        constexpr double k1p = 949.5, k1m = 182.4, k2p = 687.2, k2m = 39.4, k3p_ = 1899.0, k3m = 79300.0, k4p_ = 639.0, k4m = 40.0, Knai0 = 9.073,
                         Knao0 = 27.78, delta = -0.1550, Kki = 0.5, Kko = 0.3582, MgADP = 0.05, MgATP = 9.8, Kmgatp = 1.698e-7, H = 1.0e-7, eP = 4.2,
                         Khp = 1.698e-7, Knap = 224.0, Kxkur = 292.0;
        const double Knai = Knai0 * exp_b(delta * vfrt * (1.0 / 3.0)), Knao = Knao0 * exp_b((1.0 - delta) * vfrt * (1.0 / 3.0));
        const double Pp = eP * rcp_b(1.0 + H / Khp + nai * (1.0 / Knap) + ki * (1.0 / Kxkur));
        const double rni = nai * rcp_b(Knai), rno = nao * rcp_b(Knao), rki = ki * (1.0 / Kki), rko = ko * (1.0 / Kko);
        const double c3 = [](double x) { return x * x * x; }(1.0 + rni), c3o = [](double x) { return x * x * x; }(1.0 + rno);
        const double idni = rcp_b(c3 + (1.0 + rki) * (1.0 + rki) - 1.0), idno = rcp_b(c3o + (1.0 + rko) * (1.0 + rko) - 1.0);
        const double a1 = k1p * rni * rni * rni * idni, b1 = k1m * MgADP, a2 = k2p, b2 = k2m * rno * rno * rno * idno;
        const double a3 = k3p_ * rko * rko * idno, b3 = (k3m * Pp * H) * (1.0 / (1.0 + MgATP / Kmgatp));
        constexpr double a4 = (k4p_ * MgATP / Kmgatp) / (1.0 + MgATP / Kmgatp);
        const double b4 = k4m * rki * rki * idni;
        const double y1 = a4 * a1 * a2 + b2 * b4 * b3 + a2 * b4 * b3 + b3 * a1 * a2, y2 = b2 * b1 * b4 + a1 * a2 * a3 + a3 * b1 * b4 + a2 * a3 * b4;
        const double y3 = a2 * a3 * a4 + b3 * b2 * b1 + b2 * b1 * a4 + a3 * a4 * b1, y4 = b4 * b3 * b2 + a3 * a4 * a1 + b2 * a4 * a1 + b3 * b2 * a1;
        const double iys = rcp_b(y1 + y2 + y3 + y4), F1 = y1 * iys, F2 = y2 * iys, F3 = y3 * iys, F4 = y4 * iys;
        const double JnakNa = 3.0 * (F1 * a3 - F2 * b3), JnakK = 2.0 * (F4 * b1 - F3 * a1);
        const double INaK = 30.0 * p[8] * (celltype == 1 ? 0.9 : celltype == 2 ? 0.7 : 1.0) * (JnakNa + JnakK);
        // background currents, sarcolemmal calcium pump
        const double xkb = sg(-(v - 14.48) * (1.0 / 18.34));
        const double IKb = 0.003 * p[9] * (celltype == 1 ? 0.6 : 1.0) * xkb * (v - EK);
        const double INab = 3.75e-10 * p[10] * vffrt * (nai * e1 - nao) * ie1;
        const double ICab = 2.5e-8 * p[11] * 4.0 * vffrt * (cai * e2 - 0.341 * cao) * ie2;
        const double IpCa = 0.0005 * p[12] * cai * rcp_b(0.0005 + cai);
        du[0] = -(INa + INaL + Ito + ICaL + ICaNa + ICaK + IKr + IKs + IK1 + INaCa_i + INaCa_ss + INaK + INab + IKb + IpCa + ICab);
        // fluxes
        const double JdiffNa = (nass - nai) * 0.5, JdiffK = (kss - ki) * 0.5, Jdiff = (cass - cai) * (1.0 / 0.2);
        constexpr double bt = 4.75, a_rel = 0.5 * bt, btp = 1.25 * bt, a_relp = 0.5 * btp;
        const double r15 = 1.5 * rcp_b(cajsr), r2 = r15 * r15, r4 = r2 * r2;
        const double rel_gate = rcp_b(1.0 + r4 * r4) * (celltype == 2 ? 1.7 : 1.0);
        const double tcj = cajsr * rcp_b(cajsr + 0.0123); // 1/(1 + 0.0123/cajsr)
        const double tau_rel = fmax(bt * tcj, 0.001), tau_relp = fmax(btp * tcj, 0.001);
        du[38] = (a_rel * (-ICaL) * rel_gate - Jrelnp) * rcp_b(tau_rel);
        du[39] = (a_relp * (-ICaL) * rel_gate - Jrelp) * rcp_b(tau_relp);
        const double Jrel = fnp * Jrelnp + fpa * Jrelp;
        const double upf = celltype == 1 ? 1.3 : 1.0;
        const double Jupnp = upf * 0.004375 * cai * rcp_b(cai + 0.00092), Jupp = upf * 2.75 * 0.004375 * cai * rcp_b(cai + 0.00092 - 0.00017);
        const double Jleak = 0.0039375 * cansr * (1.0 / 15.0);
        const double Jup = fnp * Jupnp + fpa * Jupp - Jleak;
        const double Jtr = (cansr - cajsr) * (1.0 / 100.0);
        // concentrations
        du[1] = -(INa + INaL + 3.0 * INaCa_i + 3.0 * INaK + INab) * (Acap / (F * vmyo)) + JdiffNa * (vss / vmyo);
        du[2] = -(ICaNa + 3.0 * INaCa_ss) * (Acap / (F * vss)) - JdiffNa;
        du[3] = -(Ito + IKr + IKs + IK1 + IKb - 2.0 * INaK) * (Acap / (F * vmyo)) + JdiffK * (vss / vmyo);
        du[4] = -ICaK * (Acap / (F * vss)) - JdiffK;
        const double cmdnmax = celltype == 1 ? 0.05 * 1.3 : 0.05;
        constexpr double kmcmdn = 0.00238, trpnmax = 0.07, kmtrpn = 0.0005, BSRmax = 0.047, KmBSR = 0.00087, BSLmax = 1.124, KmBSL = 0.0087, csqnmax = 10.0,
                         kmcsqn = 0.8;
        const double sc1 = (kmcmdn + cai) * (kmcmdn + cai), sc2 = (kmtrpn + cai) * (kmtrpn + cai);
        const double Bcai = rcp_b(1.0 + cmdnmax * kmcmdn * rcp_b(sc1) + trpnmax * kmtrpn * rcp_b(sc2));
        du[5] = Bcai * (-(IpCa + ICab - 2.0 * INaCa_i) * (Acap / (2.0 * F * vmyo)) - Jup * (vnsr / vmyo) + Jdiff * (vss / vmyo));
        const double ss1 = (KmBSR + cass) * (KmBSR + cass), ss2 = (KmBSL + cass) * (KmBSL + cass);
        const double Bcass = rcp_b(1.0 + BSRmax * KmBSR * rcp_b(ss1) + BSLmax * KmBSL * rcp_b(ss2));
        du[6] = Bcass * (-(ICaL - 2.0 * INaCa_ss) * (Acap / (2.0 * F * vss)) + Jrel * (vjsr / vss) - Jdiff);
        du[7] = Jup - Jtr * (vjsr / vnsr);
        const double sj = (kmcsqn + cajsr) * (kmcsqn + cajsr);
        du[8] = sj * rcp_b(sj + csqnmax * kmcsqn) * (Jtr - Jrel);
    }
    template <class Emit>
    __device__ __forceinline__ static void gates(const CellParams &P, double v, Emit &&emit)
    {
        const double ko = P.p[15];
        const int celltype = (int)P.p[16];
        const double mss = sg(-(v + 39.57) * (1.0 / 9.871));
        const double tm_r = 6.765 * exp_b((v + 11.64) * (1.0 / 34.77)) + 8.552 * exp_b(-(v + 77.42) * (1.0 / 5.955)); // 1/τ_m
        const double hss = sg((v + 82.90) * (1.0 / 6.086));
        const double thf_r = 1.432e-5 * exp_b(-(v + 1.196) * (1.0 / 6.285)) + 6.149 * exp_b((v + 0.5096) * (1.0 / 20.27));
        const double ths_r = 0.009794 * exp_b(-(v + 17.95) * (1.0 / 28.05)) + 0.3343 * exp_b((v + 5.730) * (1.0 / 56.66));
        const double tj = 2.038 + rcp_b(0.02136 * exp_b(-(v + 100.6) * (1.0 / 8.281)) + 0.3052 * exp_b((v + 0.9941) * (1.0 / 38.45)));
        const double hssp = sg((v + 89.1) * (1.0 / 6.086));
        const double mLss = sg(-(v + 42.85) * (1.0 / 5.264));
        const double hLss = sg((v + 87.61) * (1.0 / 7.488)), hLssp = sg((v + 93.81) * (1.0 / 7.488));
        const double ass = sg(-(v - 14.34) * (1.0 / 14.82));
        const double ta = 1.0515 * rcp_b(sg(-(v - 18.4099) * (1.0 / 29.3814)) * (1.0 / 1.2089) + 3.5 * sg((v + 100.0) * (1.0 / 29.3814)));
        const double iss = sg((v + 43.94) * (1.0 / 5.711));
        const double delta_epi = celltype == 1 ? 1.0 - 0.95 * sg((v + 70.0) * (1.0 / 5.0)) : 1.0;
        const double tiF = (4.562 + rcp_b(0.3933 * exp_b(-(v + 100.0) * (1.0 / 100.0)) + 0.08004 * exp_b((v + 50.0) * (1.0 / 16.59)))) * delta_epi;
        const double tiS = (23.62 + rcp_b(0.001416 * exp_b(-(v + 96.52) * (1.0 / 59.05)) + 1.780e-8 * exp_b((v + 114.1) * (1.0 / 8.079)))) * delta_epi;
        const double assp = sg(-(v - 24.34) * (1.0 / 14.82));
        const double dti_develop = 1.354 + 1.0e-4 * rcp_b(exp_b((v - 167.4) * (1.0 / 15.89)) + exp_b(-(v - 12.23) * (1.0 / 0.2154)));
        const double dti_recover = 1.0 - 0.5 * sg((v + 70.0) * (1.0 / 20.0));
        const double tiFp = dti_develop * dti_recover * tiF, tiSp = dti_develop * dti_recover * tiS;
        const double dss = sg(-(v + 3.940) * (1.0 / 4.230));
        const double td = 0.6 + rcp_b(exp_b(-0.05 * (v + 6.0)) + exp_b(0.09 * (v + 14.0)));
        const double fss = sg((v + 19.58) * (1.0 / 3.696));
        const double e20 = exp_b((v + 20.0) * (1.0 / 10.0));
        const double tff = 7.0 + rcp_b(0.0045 * rcp_b(e20) + 0.0045 * e20);
        const double tfs = 1000.0 + rcp_b(0.000035 * exp_b(-(v + 5.0) * (1.0 / 4.0)) + 0.000035 * exp_b((v + 5.0) * (1.0 / 6.0)));
        const double e4 = exp_b((v - 4.0) * (1.0 / 7.0));
        const double tfcaf = 7.0 + rcp_b(0.04 * rcp_b(e4) + 0.04 * e4);
        const double tfcas = 100.0 + rcp_b(0.00012 * exp_b(-v * (1.0 / 3.0)) + 0.00012 * exp_b(v * (1.0 / 7.0)));
        const double xrss = sg(-(v + 8.337) * (1.0 / 6.789));
        const double txrf = 12.98 + rcp_b(0.3652 * exp_b((v - 31.66) * (1.0 / 3.869)) + 4.123e-5 * exp_b(-(v - 47.78) * (1.0 / 20.38)));
        const double txrs = 1.865 + rcp_b(0.06629 * exp_b((v - 34.70) * (1.0 / 7.355)) + 1.128e-5 * exp_b(-(v - 29.74) * (1.0 / 25.94)));
        const double xs1ss = sg(-(v + 11.60) * (1.0 / 8.932));
        const double txs1 = 817.3 + rcp_b(2.326e-4 * exp_b((v + 48.28) * (1.0 / 17.80)) + 0.001292 * exp_b(-(v + 210.0) * (1.0 / 230.0)));
        const double txs2_r = 0.01 * exp_b((v - 50.0) * (1.0 / 20.0)) + 0.0193 * exp_b(-(v + 66.54) * (1.0 / 31.0));
        const double xk1ss = sg(-(v + 2.5538 * ko + 144.59) / (1.5692 * ko + 3.8115));
        const double txk1_r = (exp_b(-(v + 127.2) * (1.0 / 20.36)) + exp_b((v + 236.8) * (1.0 / 69.33))) * (1.0 / 122.2);
        // rate = 1/τ
        emit(9, mss, tm_r); emit(10, hss, thf_r); emit(11, hss, ths_r);
        const double rtj = rcp_b(tj);
        emit(12, hss, rtj); emit(13, hssp, ths_r * (1.0 / 3.0)); emit(14, hss, rtj * (1.0 / 1.46));
        emit(15, mLss, tm_r); emit(16, hLss, 1.0 / 200.0); emit(17, hLssp, 1.0 / 600.0);
        const double rta = rcp_b(ta);
        emit(18, ass, rta); emit(19, iss, rcp_b(tiF)); emit(20, iss, rcp_b(tiS)); emit(21, assp, rta); emit(22, iss, rcp_b(tiFp)); emit(23, iss, rcp_b(tiSp));
        emit(24, dss, rcp_b(td)); emit(25, fss, rcp_b(tff)); emit(26, fss, rcp_b(tfs)); emit(27, fss, rcp_b(tfcaf)); emit(28, fss, rcp_b(tfcas));
        emit(29, fss, 1.0 / 75.0); emit(31, fss, rcp_b(tff) * (1.0 / 2.5)); emit(32, fss, rcp_b(tfcaf) * (1.0 / 2.5));
        emit(33, xrss, rcp_b(txrf)); emit(34, xrss, rcp_b(txrs)); emit(35, xs1ss, rcp_b(txs1)); emit(36, xs1ss, txs2_r); emit(37, xk1ss, txk1_r);
    }
    __device__ __forceinline__ static bool is_gate(int k) { return (k >= 9 && k <= 29) || (k >= 31 && k <= 37); }
    __device__ __forceinline__ static void rhs_rates(const CellParams &P, const double (&u)[NS], double, double (&du)[NS], double (&rate)[NS])
    {
#pragma unroll
        for (int k = 0; k < NS; ++k) rate[k] = 0.0;
        currents(P, u, du);
        gates(P, u[0], [&](int k, double inf, double r) { rate[k] = r; du[k] = (inf - u[k]) * r; });
    }
};

template <class M, class = void> struct uses_x : std::false_type {};
template <class M> struct uses_x<M, std::enable_if_t<M::USES_X>> : std::true_type {};

// cell_rhs!(du, u, x, t, p) (partitioned_solver.jl:88-92): x = the point's coordinate (Vec{sdim, Float32}, coordinate_systems.jl:43-49) for the
// models that read it, nothing for the others
template <class M>
__device__ __forceinline__ void cell_rhs(const CellParams &P, const double (&u)[M::NS], const float (&x)[3], double t, double (&du)[M::NS])
{
    if constexpr (uses_x<M>::value) M::rhs_x(P, u, x, t, du);
    else M::rhs(P, u, t, du);
}

// TS: storage type of the states (double, or float for the Float32 value type of the boundary: states read and rounded once per call, arithmetic in
// Float64 — the same result as converting, stepping and converting back, in one pass over half the bytes)
template <int MODEL, int LAYOUT, bool WRITE_DU, class TS = double>
__global__ void __launch_bounds__(256)
k_reaction(CellParams P, TS *__restrict__ u, TS *__restrict__ du_out, int64_t n, double t, double dt, int substeps,
           double threshold, unsigned long long *__restrict__ rmax_key, const float *__restrict__ xs, int sdim, const double *__restrict__ tslot)
{
    if (tslot) t = tslot[0]; // replayed from a graph: the time of this launch sits on the device (tb_graph.hip)
    using M = CellModel<MODEL>;
    constexpr int NS = M::NS;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    double rm = -__builtin_huge_val(); // reaction tangent: max of the φₘ rate of the last evaluation (rtc.jl:64-73)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double ul[NS], dul[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) ul[j] = LAYOUT == TB_LAYOUT_SOA ? u[i + j * n] : u[i * NS + j];
        float xp[3] = {0.0f, 0.0f, 0.0f};
        if constexpr (uses_x<M>::value) {
            if (xs) for (int d = 0; d < sdim; ++d) xp[d] = xs[i * sdim + d];
        }
        cell_rhs<M>(P, ul, xp, t, dul);
        if (substeps <= 1 || fabs(dul[M::PHI]) < threshold) {
#pragma unroll
            for (int j = 0; j < NS; ++j) ul[j] += dt * dul[j];
        } else {
            const double dts = dt / substeps;
#pragma unroll
            for (int j = 0; j < NS; ++j) ul[j] += dts * dul[j];
            for (int s = 2; s <= substeps; ++s) {
                cell_rhs<M>(P, ul, xp, t + (s - 1) * dts, dul);
#pragma unroll
                for (int j = 0; j < NS; ++j) ul[j] += dts * dul[j];
            }
        }
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const int64_t k = LAYOUT == TB_LAYOUT_SOA ? i + j * n : i * NS + j;
            u[k] = (TS)ul[j];
            if (WRITE_DU) du_out[k] = (TS)dul[j];
        }
        rm = fmax(rm, dul[M::PHI]);
    }
    if (rmax_key) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) rm = fmax(rm, __shfl_xor(rm, o, 64));
        if ((threadIdx.x & 63) == 0) {
            const unsigned long long b = (unsigned long long)__double_as_longlong(rm);
            atomicMax(rmax_key, (b >> 63) ? ~b : (b | 0x8000000000000000ull)); // order-preserving key, see tb_algebra.hip
        }
    }
}

// expm1 for the Rush–Larsen factor: series for small arguments (no cancellation), exp − 1 otherwise
__device__ __forceinline__ double expm1_b(double z)
{
    if (fabs(z) < 0.3) {
        double q = 1.6059043836821613e-10;
        const double c[11] = {2.08767569878681e-09, 2.505210838544172e-08, 2.755731922398589e-07, 2.7557319223985893e-06, 2.48015873015873e-05,
                              0.0001984126984126984, 0.001388888888888889, 0.008333333333333333, 0.041666666666666664, 0.16666666666666666, 0.5};
#pragma unroll
        for (int i = 0; i < 11; ++i) q = fma(q, z, c[i]);
        q = fma(q, z, 1.0);
        return q * z;
    }
    return exp_b(z) - 1.0;
}

// Rush–Larsen step (the splitting SURVEY §8 f4 names; the reference only carries the reaction_rhs!/state_rhs! hooks for it,
// src/modeling/cells/fhn.jl:36-60): gates are advanced with the exact solution of their linear ODE for frozen V,
//   y ← y∞ + (y − y∞) e^{−Δt/τ}  =  y + du · (1 − e^{−Δt·rate}) / rate,
// every other state by forward Euler.  Removes the fast sodium gate's µs-scale stability limit: TT06 runs at Δt = 0.02 ms
// in one evaluation instead of twenty forward-Euler sub-steps.
template <int MODEL, int LAYOUT>
__global__ void __launch_bounds__(256)
k_reaction_rl(CellParams P, double *__restrict__ u, int64_t n, double t, double dt, const double *__restrict__ tslot)
{
    if (tslot) t = tslot[0];
    using M = CellModel<MODEL>;
    constexpr int NS = M::NS;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double ul[NS], dul[NS], rate[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) ul[j] = LAYOUT == TB_LAYOUT_SOA ? u[i + j * n] : u[i * NS + j];
        M::rhs_rates(P, ul, t, dul, rate);
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const double h = rate[j] != 0.0 ? -expm1_b(-dt * rate[j]) * rcp_b(rate[j]) : dt;
            const int64_t k = LAYOUT == TB_LAYOUT_SOA ? i + j * n : i * NS + j;
            u[k] = ul[j] + h * dul[j];
        }
    }
}

int launch_reaction_rl(tb_device *dev, int model, const double *params, int n_params, double *d_u, int64_t n, int layout, double t, double dt)
{
    if (model != TB_CELL_TT06 && model != TB_CELL_PCG2019 && model != TB_CELL_ORD11) {
        set_error("Rush–Larsen step: cell model %d has no gate decomposition (TB_CELL_TT06, TB_CELL_PCG2019 and TB_CELL_ORD11 have)", model);
        return TB_ERR_UNSUPPORTED;
    }
    CellParams P{};
    for (int i = 0; i < n_params && i < 48; ++i) P.p[i] = params[i];
    int64_t nb = (n + 255) / 256;
    const int64_t cap = (int64_t)dev->n_cu * 16;
    if (nb > cap) nb = (nb + (nb + cap - 1) / cap - 1) / ((nb + cap - 1) / cap); // balanced: every thread ⌈nb / cap⌉ points
    if (n == 0) return TB_OK;
    const double *ts = dev->capturing ? (dev->tslot_used = true, dev->d_tslot) : nullptr;
#define TB_RL(MODEL) do { if (layout == TB_LAYOUT_SOA) hipLaunchKernelGGL((k_reaction_rl<MODEL, TB_LAYOUT_SOA>), dim3((unsigned)nb), dim3(256), 0, dev->stream, P, d_u, n, t, dt, ts); \
                          else hipLaunchKernelGGL((k_reaction_rl<MODEL, TB_LAYOUT_AOS>), dim3((unsigned)nb), dim3(256), 0, dev->stream, P, d_u, n, t, dt, ts); } while (0)
    if (model == TB_CELL_TT06) TB_RL(TB_CELL_TT06); else if (model == TB_CELL_ORD11) TB_RL(TB_CELL_ORD11); else TB_RL(TB_CELL_PCG2019);
#undef TB_RL
    TB_HIP(hipGetLastError());
    return TB_OK;
}

template <int MODEL, class TS = double>
static int run(tb_device *dev, const CellParams &P, TS *u, TS *du, int64_t n, int layout, double t, double dt, int substeps,
               double thr, unsigned long long *rmax_key, const float *xs, int sdim)
{
    const int bs = 256;
    // Grid: a multiple of what is resident (`occ` workgroups per CU: 2 for the 19-state model at 243 registers, 8 for the two-state ones), every thread the
    // same number of points.  Two generations of resident workgroups for large arrays, ONE for small ones (≤ 12 generations of single-point workgroups: the
    // 27-layer slab of a strong-scaling run) — measured at steady clocks, TT06 (profiles/r05_v2/ab_reaction_grid.log): 216³ 0.715 ms against 0.721 at
    // 16 workgroups per CU; 54 / 27 layers 74.2 / 69.5 ps per dof against 78.0 / 74.5.
    auto launch = [&](auto kern) -> int {
        static struct { const void *f; int occ; } seen[8]; // (the query is not free: once per kernel instance)
        int occ = 0;
        for (auto &e : seen) if (e.f == (const void *)kern) occ = e.occ;
        if (!occ) {
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, bs, 0) != hipSuccess || occ < 1) occ = 1;
            for (auto &e : seen) if (!e.f) { e.f = (const void *)kern; e.occ = occ; break; }
        }
        int64_t nb = (n + bs - 1) / bs;
        static const int per_cu_env = tune_env("TB_REACTION_BLOCKS_PER_CU") ? atoi(tune_env("TB_REACTION_BLOCKS_PER_CU")) : -1;
        const int64_t resident = (int64_t)dev->n_cu * occ;
        const int64_t cap = per_cu_env > 0 ? (int64_t)dev->n_cu * per_cu_env : per_cu_env == 0 ? nb : (nb <= 12 * resident ? resident : 2 * resident);
        if (nb > cap) nb = (nb + (nb + cap - 1) / cap - 1) / ((nb + cap - 1) / cap); // every thread the same number of points (⌈nb / cap⌉ each)
        hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(bs), 0, dev->stream, P, u, du, n, t, dt, substeps, thr, rmax_key, xs, sdim,
                           dev->capturing ? (dev->tslot_used = true, (const double *)dev->d_tslot) : (const double *)nullptr);
        return TB_OK;
    };
    if (layout == TB_LAYOUT_SOA) { if (du) launch(k_reaction<MODEL, TB_LAYOUT_SOA, true, TS>); else launch(k_reaction<MODEL, TB_LAYOUT_SOA, false, TS>); }
    else { if (du) launch(k_reaction<MODEL, TB_LAYOUT_AOS, true, TS>); else launch(k_reaction<MODEL, TB_LAYOUT_AOS, false, TS>); }
    TB_HIP(hipGetLastError());
    return TB_OK;
}

// Float32 storage (tb_reaction_step_f32): same kernels with TS = float
int launch_reaction_f32(tb_device *dev, int model, const double *params, int n_params, float *d_u, float *d_du, int64_t n_points,
                        int layout, double t, double dt, int substeps, double thr, const float *d_x, int sdim)
{
    CellParams P{};
    for (int i = 0; i < n_params && i < 48; ++i) P.p[i] = params[i];
    switch (model) {
    case TB_CELL_FHN: return run<TB_CELL_FHN, float>(dev, P, d_u, d_du, n_points, layout, t, dt, substeps, thr, nullptr, d_x, sdim);
    case TB_CELL_ALIEV_PANFILOV: return run<TB_CELL_ALIEV_PANFILOV, float>(dev, P, d_u, d_du, n_points, layout, t, dt, substeps, thr, nullptr, d_x, sdim);
    case TB_CELL_PCG2019: return run<TB_CELL_PCG2019, float>(dev, P, d_u, d_du, n_points, layout, t, dt, substeps, thr, nullptr, d_x, sdim);
    case TB_CELL_TT06: return run<TB_CELL_TT06, float>(dev, P, d_u, d_du, n_points, layout, t, dt, substeps, thr, nullptr, d_x, sdim);
    case TB_CELL_FHN_HETEROGENEOUS: return run<TB_CELL_FHN_HETEROGENEOUS, float>(dev, P, d_u, d_du, n_points, layout, t, dt, substeps, thr, nullptr, d_x, sdim);
    case TB_CELL_ORD11: return run<TB_CELL_ORD11, float>(dev, P, d_u, d_du, n_points, layout, t, dt, substeps, thr, nullptr, d_x, sdim);
    default: set_error("unknown cell model %d", model); return TB_ERR_BAD_ARG;
    }
}

int launch_reaction(tb_device *dev, int model, const double *params, int n_params, double *d_u, double *d_du, int64_t n_points,
                    int layout, double t, double dt, int substeps, double thr, double *rmax, const float *d_x, int sdim)
{
    CellParams P{};
    for (int i = 0; i < n_params && i < 48; ++i) P.p[i] = params[i];
    unsigned long long *key = nullptr;
    if (rmax) {
        TB_NO_CAPTURE(dev); // the largest rate goes to the host
        key = (unsigned long long *)&dev->d_status->cell; // 8-byte scratch inside the status block
        TB_HIP(hipMemsetAsync(key, 0, sizeof *key, dev->stream));
    }
    int rc = TB_ERR_BAD_ARG;
    switch (model) {
    case TB_CELL_FHN: rc = run<TB_CELL_FHN>(dev, P, d_u, d_du, n_points, layout, t, dt, substeps, thr, key, d_x, sdim); break;
    case TB_CELL_ALIEV_PANFILOV: rc = run<TB_CELL_ALIEV_PANFILOV>(dev, P, d_u, d_du, n_points, layout, t, dt, substeps, thr, key, d_x, sdim); break;
    case TB_CELL_PCG2019: rc = run<TB_CELL_PCG2019>(dev, P, d_u, d_du, n_points, layout, t, dt, substeps, thr, key, d_x, sdim); break;
    case TB_CELL_TT06: rc = run<TB_CELL_TT06>(dev, P, d_u, d_du, n_points, layout, t, dt, substeps, thr, key, d_x, sdim); break;
    case TB_CELL_FHN_HETEROGENEOUS: rc = run<TB_CELL_FHN_HETEROGENEOUS>(dev, P, d_u, d_du, n_points, layout, t, dt, substeps, thr, key, d_x, sdim); break;
    case TB_CELL_ORD11: rc = run<TB_CELL_ORD11>(dev, P, d_u, d_du, n_points, layout, t, dt, substeps, thr, key, d_x, sdim); break;
    default: set_error("unknown cell model %d", model); return TB_ERR_BAD_ARG;
    }
    if (rc || !rmax) return rc;
    unsigned long long bits = 0;
    TB_HIP(hipMemcpyAsync(&bits, key, sizeof bits, hipMemcpyDeviceToHost, dev->stream));
    TB_SYNC_STREAM(dev);
    *rmax = bits ? decode_ordered_key(bits) : -__builtin_huge_val();
    return TB_OK;
}

} // namespace tb
