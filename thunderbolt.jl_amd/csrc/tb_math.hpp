// tb_math.hpp — bounded-argument exponential and refined reciprocal shared by the reaction and the source-term kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace tb {

// workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, i.e. waits for every global store and prefetch load in flight
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// exp for bounded arguments (|x| ≲ 700; ionic-model arguments stay within ±100): k = rint(x·log₂e), r = x − k·ln2 in two
// pieces (|r| ≤ 0.347), degree-13 Taylor polynomial (truncation 2·10⁻¹⁸ relative), one ldexp.  ≈21 instructions against ≈40 of
// the library routine, which spends the rest on special cases that cannot occur here; agreement with libm ≤ 2 ulp
// (tests/test_gpu_parity.py compares whole trajectories at 1e-12).  The reaction kernels are exp-bound, not HBM-bound.
__device__ __forceinline__ double exp_b(double x)
{
    x = fmin(fmax(x, -700.0), 700.0);
    const double kf = rint(x * 1.4426950408889634);
    double r = fma(kf, -6.93147180369123816490e-01, x);
    r = fma(kf, -1.90821492927058770002e-10, r);
    const double c[12] = {1.6059043836821613e-10, 2.08767569878681e-09, 2.505210838544172e-08, 2.755731922398589e-07, 2.7557319223985893e-06, 2.48015873015873e-05, 0.0001984126984126984, 0.001388888888888889, 0.008333333333333333, 0.041666666666666664, 0.16666666666666666, 0.5};
    double q = c[0];
#pragma unroll
    for (int i = 1; i < 12; ++i) q = fma(q, r, c[i]);
    q = fma(q, r, 1.0); // … + r
    q = fma(q, r, 1.0); // 1 + r·(…)
    return ldexp(q, (int)kf);
}

// 1/y for well-scaled arguments: hardware reciprocal refined by two Newton steps (≤ 1–2 ulp), 5 instructions against the
// ≈12 of the IEEE division sequence (no scaling / fix-up: gate and buffer denominators are O(1) numbers)
__device__ __forceinline__ double rcp_b(double y)
{
    double r = __builtin_amdgcn_rcp(y);
    r = fma(fma(-y, r, 1.0), r, r);
    r = fma(fma(-y, r, 1.0), r, r);
    return r;
}

// log for positive normal arguments (concentrations, 10⁻⁷ … 10³): x = m·2ᵉ with m ∈ [√½, √2), log m = 2 atanh f, f = (m − 1)/(m + 1), |f| ≤ 0.172,
// odd series to f²¹ (truncation 4·10⁻¹⁸).  ≈ 27 instructions; no special cases (zero, negative, subnormal, infinite arguments cannot occur)
__device__ __forceinline__ double log_b(double x)
{
    int e;
    double m = frexp(x, &e); // [0.5, 1)
    if (m < 0.70710678118654752) { m += m; --e; }
    const double f = (m - 1.0) * rcp_b(m + 1.0), f2 = f * f;
    const double c[10] = {1.0 / 21.0, 1.0 / 19.0, 1.0 / 17.0, 1.0 / 15.0, 1.0 / 13.0, 1.0 / 11.0, 1.0 / 9.0, 1.0 / 7.0, 1.0 / 5.0, 1.0 / 3.0};
    double q = c[0];
#pragma unroll
    for (int i = 1; i < 10; ++i) q = fma(q, f2, c[i]);
    q = fma(q * f2, f, f); // f + f³(…)
    const double ef = (double)e;
    return fma(ef, 6.93147180369123816490e-01, fma(ef, 1.90821492927058770002e-10, q + q));
}

} // namespace tb
