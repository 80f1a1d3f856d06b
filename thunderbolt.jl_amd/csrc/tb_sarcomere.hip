// Pointwise sarcomere dynamics on the device: the StandaloneSarcomereModel protocol of the reference
// (src/modeling/solid/contraction.jl:150-163: du = sarcomere_rhs!(u, λ(t), dλdt(t), Ca(t))) advanced by forward Euler, one lane per
// point, the 20 states of a point in registers, states stored point-fastest (SoA) like the cell models.
// HBM-bound: 2·20·8 B state traffic + up to 24 B inputs + 16 B outputs per point and launch.
#include "tb_internal.h"
#include "tb_sarcomere.hpp"

namespace tb {
using namespace tbk;

struct SarcomereInputs {
    const double *stretch, *velocity, *calcium; // per point, or nullptr → the scalar below
    double s_stretch, s_velocity, s_calcium;
};

__global__ void __launch_bounds__(256)
k_sarcomere_rdq20(RDQ20Params p, SarcomereInputs in, double *__restrict__ state, int64_t n, double dt, int substeps, int rate_independent,
                  double *__restrict__ tension, double *__restrict__ stiffness)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double u[RDQ20_NSTATES], du[RDQ20_NSTATES];
#pragma unroll
        for (int k = 0; k < RDQ20_NSTATES; ++k) u[k] = state[(int64_t)k * n + i];
        const double lam = in.stretch ? in.stretch[i] : in.s_stretch;
        const double vel = rate_independent ? 0.0 : (in.velocity ? in.velocity[i] : in.s_velocity);
        const double ca = in.calcium ? in.calcium[i] : in.s_calcium;
        for (int s = 0; s < substeps; ++s) {
            rdq20_rhs(p, u, lam, vel, ca, du);
#pragma unroll
            for (int k = 0; k < RDQ20_NSTATES; ++k) u[k] += dt * du[k];
        }
#pragma unroll
        for (int k = 0; k < RDQ20_NSTATES; ++k) state[(int64_t)k * n + i] = u[k];
        if (tension) tension[i] = rdq20_tension(p, u, lam);
        if (stiffness) stiffness[i] = rdq20_stiffness(p, u, lam);
    }
}

static RDQ20Params rdq20_params(const double *q)
{
    RDQ20Params p;
    p.LA = q[0]; p.LM = q[1]; p.LB = q[2]; p.SL0 = q[3]; p.Q = q[4]; p.Kd0 = q[5]; p.aKd = q[6]; p.mu = q[7]; p.gamma = q[8]; p.Koff = q[9];
    p.Kbasic = q[10]; p.r0 = q[11]; p.alpha = q[12]; p.mu0_fP = q[13]; p.mu1_fP = q[14]; p.a_XB = q[15]; p.eps_v = q[16];
    return p;
}

int launch_sarcomere(tb_device *dev, const double *params, double *d_state, int64_t n, const double *d_stretch, const double *d_velocity,
                     const double *d_calcium, double stretch, double velocity, double calcium, double dt, int substeps, int rate_independent,
                     double *d_tension, double *d_stiffness)
{
    const SarcomereInputs in{d_stretch, d_velocity, d_calcium, stretch, velocity, calcium};
    int64_t nb = (n + 255) / 256;
    const int64_t cap = (int64_t)dev->n_cu * 16;
    if (nb > cap) nb = cap;
    hipLaunchKernelGGL(k_sarcomere_rdq20, dim3((unsigned)nb), dim3(256), 0, dev->stream, rdq20_params(params), in, d_state, n, dt, substeps,
                       rate_independent, d_tension, d_stiffness);
    TB_HIP(hipGetLastError());
    return TB_OK;
}

void host_sarcomere_eval(const double *params, const double *u, double stretch, double velocity, double calcium, double *du, double *tension,
                         double *stiffness)
{
    const RDQ20Params p = rdq20_params(params);
    double ul[RDQ20_NSTATES], dul[RDQ20_NSTATES];
    for (int k = 0; k < RDQ20_NSTATES; ++k) ul[k] = u[k];
    rdq20_rhs(p, ul, stretch, velocity, calcium, dul);
    if (du) for (int k = 0; k < RDQ20_NSTATES; ++k) du[k] = dul[k];
    if (tension) *tension = rdq20_tension(p, ul, stretch);
    if (stiffness) *stiffness = rdq20_stiffness(p, ul, stretch);
}

} // namespace tb
