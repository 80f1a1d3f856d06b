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

#include "tb_internal.h"

namespace tb {

struct CellParams {
    double p[36];
};

template <int MODEL> struct CellModel;

template <> struct CellModel<TB_CELL_FHN> {
    static constexpr int NS = 2, PHI = 0;
    __device__ __forceinline__ static void rhs(const CellParams &P, const double (&u)[NS], double, double (&du)[NS])
    {
        const double a = P.p[0], b = P.p[1], c = P.p[2], d = P.p[3], e = P.p[4], f = P.p[5];
        const double phi = u[0], s = u[1];
        du[0] = f * (phi * (1 - phi) * (phi - a) - s);
        du[1] = e * (b * phi - c * s - d);
    }
};

template <> struct CellModel<TB_CELL_ALIEV_PANFILOV> {
    static constexpr int NS = 2, PHI = 1; // state order (s, φₘ)
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
    // parameter slots follow the struct field order of pcg2019.jl:4-48
    enum { gNa, Em, km, taum, Eh, kh, dh, tauh0, gK1, Ez, kz, gto, Er, kr, Es, ks, taus, gCaL, Ed, kd, Ef, kf, tauf,
           gKr, Exr, kxr, tauxr, Ey, ky, gKs, Exs, kxs, tauxs, ENa, EK, ECa };
    __device__ __forceinline__ static double sigmoid(double phi, double E, double k, double sign)
    {
        return 1.0 / (1.0 + exp(sign * (phi - E) / k));
    }
    __device__ __forceinline__ static void rhs(const CellParams &P, const double (&u)[NS], double, double (&du)[NS])
    {
        const double *p = P.p;
        const double phi = u[0], h = u[1], m = u[2], f = u[3], s = u[4], xs = u[5], xr = u[6];
        const double rinf = sigmoid(phi, p[Er], p[kr], -1.0);
        const double dinf = sigmoid(phi, p[Ed], p[kd], -1.0);
        const double zinf = sigmoid(phi, p[Ez], p[kz], 1.0);
        const double yinf = sigmoid(phi, p[Ey], p[ky], 1.0);
        const double I_Na = p[gNa] * m * m * m * h * h * (phi - p[ENa]);
        const double I_K1 = p[gK1] * zinf * (phi - p[EK]);
        const double I_to = p[gto] * rinf * s * (phi - p[EK]);
        const double I_CaL = p[gCaL] * dinf * f * (phi - p[ECa]);
        const double I_Kr = p[gKr] * xr * yinf * (phi - p[EK]);
        const double I_Ks = p[gKs] * xs * (phi - p[EK]);
        du[0] = -(I_Na + I_K1 + I_to + I_CaL + I_Kr + I_Ks); // C_m = 1, pcg2019.jl:55
        const double eh = exp((phi - p[Eh]) / p[kh]);
        const double tau_h = (2.0 * p[tauh0] * exp(p[dh] * (phi - p[Eh]) / p[kh])) / (1.0 + eh);
        const double hinf = 1.0 / (1.0 + eh);
        du[1] = (hinf - h) / tau_h;
        du[2] = (sigmoid(phi, p[Em], p[km], -1.0) - m) / p[taum];
        du[3] = (sigmoid(phi, p[Ef], p[kf], 1.0) - f) / p[tauf];
        du[4] = (sigmoid(phi, p[Es], p[ks], 1.0) - s) / p[taus];
        du[5] = (sigmoid(phi, p[Exs], p[kxs], -1.0) - xs) / p[tauxs];
        du[6] = (sigmoid(phi, p[Exr], p[kxr], -1.0) - xr) / p[tauxr];
    }
};

template <int MODEL, int LAYOUT, bool WRITE_DU>
__global__ void __launch_bounds__(256)
k_reaction(CellParams P, double *__restrict__ u, double *__restrict__ du_out, int64_t n, double t, double dt, int substeps,
           double threshold)
{
    using M = CellModel<MODEL>;
    constexpr int NS = M::NS;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double ul[NS], dul[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) ul[j] = LAYOUT == TB_LAYOUT_SOA ? u[i + j * n] : u[i * NS + j];
        M::rhs(P, ul, t, dul);
        if (substeps <= 1 || fabs(dul[M::PHI]) < threshold) {
#pragma unroll
            for (int j = 0; j < NS; ++j) ul[j] += dt * dul[j];
        } else {
            const double dts = dt / substeps;
#pragma unroll
            for (int j = 0; j < NS; ++j) ul[j] += dts * dul[j];
            for (int s = 2; s <= substeps; ++s) {
                M::rhs(P, ul, t + (s - 1) * dts, dul);
#pragma unroll
                for (int j = 0; j < NS; ++j) ul[j] += dts * dul[j];
            }
        }
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const int64_t k = LAYOUT == TB_LAYOUT_SOA ? i + j * n : i * NS + j;
            u[k] = ul[j];
            if (WRITE_DU) du_out[k] = dul[j];
        }
    }
}

template <int MODEL>
static int run(tb_device *dev, const CellParams &P, double *u, double *du, int64_t n, int layout, double t, double dt, int substeps,
               double thr)
{
    const int bs = 256;
    int64_t nb = (n + bs - 1) / bs;
    const int64_t cap = (int64_t)dev->n_cu * 16;
    if (nb > cap) nb = cap;
    const dim3 grid((unsigned)nb), block(bs);
#define TB_LAUNCH(L, W) hipLaunchKernelGGL((k_reaction<MODEL, L, W>), grid, block, 0, dev->stream, P, u, du, n, t, dt, substeps, thr)
    if (layout == TB_LAYOUT_SOA) { if (du) TB_LAUNCH(TB_LAYOUT_SOA, true); else TB_LAUNCH(TB_LAYOUT_SOA, false); }
    else { if (du) TB_LAUNCH(TB_LAYOUT_AOS, true); else TB_LAUNCH(TB_LAYOUT_AOS, false); }
#undef TB_LAUNCH
    TB_HIP(hipGetLastError());
    return TB_OK;
}

int launch_reaction(tb_device *dev, int model, const double *params, int n_params, double *d_u, double *d_du, int64_t n_points,
                    int layout, double t, double dt, int substeps, double thr)
{
    CellParams P{};
    for (int i = 0; i < n_params && i < 36; ++i) P.p[i] = params[i];
    switch (model) {
    case TB_CELL_FHN: return run<TB_CELL_FHN>(dev, P, d_u, d_du, n_points, layout, t, dt, substeps, thr);
    case TB_CELL_ALIEV_PANFILOV: return run<TB_CELL_ALIEV_PANFILOV>(dev, P, d_u, d_du, n_points, layout, t, dt, substeps, thr);
    case TB_CELL_PCG2019: return run<TB_CELL_PCG2019>(dev, P, d_u, d_du, n_points, layout, t, dt, substeps, thr);
    }
    set_error("unknown cell model %d", model);
    return TB_ERR_BAD_ARG;
}

} // namespace tb
