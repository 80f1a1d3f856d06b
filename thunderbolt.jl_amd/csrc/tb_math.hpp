// tb_math.hpp — bounded-argument exponential and refined reciprocal shared by the reaction and the source-term kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace tb {

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

} // namespace tb
