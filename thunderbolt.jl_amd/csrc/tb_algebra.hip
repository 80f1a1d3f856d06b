// tb_algebra.hip — the heat-step algebra around the assembled operators (all HBM-bound streams):
//   Anz = Mnz − Δt·Knz                src/solver/time/euler.jl:110-116
//   y = α·A·x + β·y (CSR)             src/utils.jl:185-231  (`b = M uₙ₋₁`, euler.jl:85)
//   y += a·x                          add!(b, source), euler.jl:90
//   max |x[i·stride]|                 RTC controller input, src/solver/time/rtc.jl:64-73
#include <hip/hip_runtime.h>

#include <unordered_map>

#include <cmath>
#include <cstring>

#include "tb_internal.h"

namespace tb {

// TB_SPMV_KERNEL: the product library honours "rows" only — the CSR rows kernel, which is also what a pattern without shared row signatures runs (the
// switch lets a test put it on a compressible pattern and compare bits); the older entry-per-lane kernels ("rec", "chain") and the wave-private form
// ("wave", measured slower on thin slabs) are comparison builds: profiling library only
static const char *spmv_kernel_env()
{
    const char *e = getenv("TB_SPMV_KERNEL");
#ifndef TB_ABLATION
    if (e && strcmp(e, "rows") != 0 && strcmp(e, "sig") != 0) return nullptr;
#endif
    return e;
}

// grid of a kernel of 1 024-thread workgroups that ends in a workgroup sum (block_sum_to): two per CU
static inline unsigned grid_red(tb_device *dev, int64_t n)
{
    const int64_t nb = (n + 1023) / 1024, cap = (int64_t)dev->n_cu * 2;
    return (unsigned)std::max<int64_t>(1, nb > cap ? cap : nb);
}
static inline unsigned grid_for(tb_device *dev, int64_t n, int bs)
{
    int64_t nb = (n + bs - 1) / bs;
    const int64_t cap = (int64_t)dev->n_cu * 8;
    return (unsigned)(nb > cap ? cap : nb);
}

__global__ void __launch_bounds__(256)
k_heat_matrix(int64_t n, const double *__restrict__ M, const double *__restrict__ K, double dt, double *__restrict__ A)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t n2 = n >> 1;
    const double2 *M2 = (const double2 *)M;
    const double2 *K2 = (const double2 *)K;
    double2 *A2 = (double2 *)A;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
        const double2 m = M2[i], k = K2[i];
        A2[i] = make_double2(m.x - dt * k.x, m.y - dt * k.y);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) A[n - 1] = M[n - 1] - dt * K[n - 1];
}

__global__ void __launch_bounds__(256) k_axpy(int64_t n, double a, const double *__restrict__ x, double *__restrict__ y)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] += a * x[i];
}

// CSR SpMV, LANES lanes per row (FE rows hold ~27 nz): row-contiguous reads of nzval / colidx
template <int LANES>
__global__ void __launch_bounds__(256)
k_spmv(int64_t nrows, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ colidx, const double *__restrict__ nz,
       const double *__restrict__ x, double alpha, double beta, double *__restrict__ y)
{
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int sub = threadIdx.x % LANES;
    const int64_t nsub = ((int64_t)gridDim.x * blockDim.x) / LANES;
    for (int64_t r = gid / LANES; r < nrows; r += nsub) {
        const int64_t k0 = rowptr[r], k1 = rowptr[r + 1];
        double v = 0.0;
        for (int64_t k = k0 + sub; k < k1; k += LANES) v += nz[k] * x[colidx[k]];
#pragma unroll
        for (int o = LANES / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, LANES);
        if (sub == 0) y[r] = beta == 0.0 ? alpha * v : alpha * v + beta * y[r];
    }
}

__device__ __forceinline__ void block_sum_to(double v, double *out);
__device__ __forceinline__ void block_sum_slots(double v, double *group);

// CSR "stream" SpMV: a workgroup owns a run of consecutive rows holding ≤ CAP non-zeros.  Phase 1 streams nzval / colidx of the whole run with
// every lane busy and fully coalesced (lane i takes entry i, whatever row it belongs to), gathers x and parks the products in LDS; phase 2 sums
// each row's segment with 8 lanes.  Against the lanes-per-row kernel (27-entry rows fill 27 of 32 lane slots and issue two dependent passes)
// this keeps CAP/256 independent loads in flight per lane.  DOT: also accumulates xᵀy (the pᵀAp of CG) into *xy.
template <int CAP, bool DOT>
__global__ void __launch_bounds__(256)
k_spmv_stream(int n_blk, const int32_t *__restrict__ blkrow, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ colidx, const double *__restrict__ nz,
              const double *__restrict__ x, double alpha, double beta, double *__restrict__ y, double *__restrict__ xy)
{
    __shared__ double s[CAP];
    constexpr int LN = 8;
    const int sub = threadIdx.x % LN;
    double acc = 0.0;
    for (int b = blockIdx.x; b < n_blk; b += gridDim.x) {
        const int r0 = blkrow[b], r1 = blkrow[b + 1];
        const int64_t k0 = rowptr[r0];
        const int len = (int)(rowptr[r1] - k0);
        const double *nzb = nz + k0;
        const int32_t *cb = colidx + k0;
#pragma unroll
        for (int u = 0; u < CAP / 256; ++u) {
            const int i = threadIdx.x + u * 256;
#ifdef TB_SPMV_NT
            if (i < len) s[i] = __builtin_nontemporal_load(nzb + i) * x[__builtin_nontemporal_load(cb + i)];
#else
            if (i < len) s[i] = nzb[i] * x[cb[i]];
#endif
        }
        __syncthreads();
        for (int r = r0 + threadIdx.x / LN; r < r1; r += 256 / LN) {
            const int a = (int)(rowptr[r] - k0), e = (int)(rowptr[r + 1] - k0);
            double v = 0.0;
            for (int i = a + sub; i < e; i += LN) v += s[i];
#pragma unroll
            for (int o = LN / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, LN);
            if (sub == 0) {
                if constexpr (DOT) { y[r] = v; acc += x[r] * v; }
                else y[r] = beta == 0.0 ? alpha * v : alpha * v + beta * y[r];
            }
        }
        __syncthreads();
    }
    if constexpr (DOT) block_sum_slots(acc, xy); // xy: a slot group (block_sum_slots below)
}

#ifdef TB_ABLATION
__device__ int g_spmv_nogather = 0;
#endif
// The same kernel with the dependent trips of a block cut from five to two.  Above, a block walks block → row range → row pointers → entries → x,
// and after the barrier every row group loads its two row pointers again: with eight resident workgroups per CU the waves sit in metadata trips four
// fifths of the time.  Here a block is ONE 16-byte record {first row, rows | entries << 16, first nz} whose load for the NEXT block is issued at the top
// of the current one, and the row offsets of the first three passes of phase 2 are requested together with the entries (NPRE·32 rows: every row of
// a 27-entries-per-row block), so a block costs record (hidden) → entries + offsets → x.  What is left is the gather itself: a profiling build that reads x
// coalesced instead runs at 0.57 instead of 0.81 ms at 216³.  A windowed form (x of the block's ≈ 9 runs of consecutive columns copied into LDS, 16-bit
// window positions instead of 32-bit columns, 10 B per entry) was built and is correct, but not faster: 0.74–0.75 ms against 0.73 ms on the same box with
// register staging (run scan by readlane, LDS gather, a third barrier, four workgroups per CU), 0.98 ms with LDS-DMA staging (hipcc 7.2 follows every
// `global_load_lds` in a loop by `s_waitcnt vmcnt(0)`); removed again.
template <int CAP, bool DOT>
__global__ void __launch_bounds__(256)
k_spmv_stream_rec(int n_blk, const uint4 *__restrict__ blkrec, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ colidx, const double *__restrict__ nz,
                  const double *__restrict__ x, double alpha, double beta, double *__restrict__ y, double *__restrict__ xy)
{
    __shared__ double s[CAP];
    constexpr int LN = 8, NG = 256 / LN, NPRE = 3;
    const int sub = threadIdx.x % LN, g = threadIdx.x / LN;
    double acc = 0.0;
    int b = blockIdx.x;
    uint4 rec = blkrec[b < n_blk ? b : 0];
    for (; b < n_blk; b += gridDim.x) {
        const int bn = b + gridDim.x;
        const uint4 recn = blkrec[bn < n_blk ? bn : b];
        const int r0 = (int)rec.x, nr = (int)(rec.y & 0xffffu), len = (int)(rec.y >> 16);
        const int64_t k0 = (int64_t)(((uint64_t)rec.w << 32) | rec.z);
        // every load below is unconditional (indices clamped into the block): a load inside `if (i < len)` is followed by its own wait, which made the
        // eight entry / gather pairs of a lane sixteen trips one after the other
        int64_t pa[NPRE], pe[NPRE];
#pragma unroll
        for (int j = 0; j < NPRE; ++j) {
            const int r = g + NG * j, rc = r < nr ? r : nr - 1;
            pa[j] = rowptr[r0 + rc]; pe[j] = rowptr[r0 + rc + 1];
        }
        const double *nzb = nz + k0;
        const int32_t *cb = colidx + k0;
        constexpr int U = CAP / 256;
        int32_t cj[U];
        double vj[U], xj[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = threadIdx.x + u * 256, ic = i < len ? i : len - 1;
            cj[u] = 0; vj[u] = 0.0;
            if (len > 0) { cj[u] = cb[ic]; vj[u] = nzb[ic]; } // wave-uniform condition (a run of empty rows has no entries to read)
        }
#ifdef TB_ABLATION
        if (g_spmv_nogather) { // profiling build: what the kernel costs without the gather of x (coalesced reads of the same volume instead)
#pragma unroll
            for (int u = 0; u < U; ++u) xj[u] = x[(cj[u] & 0) + r0 + ((threadIdx.x + u * 256) & 63)];
        } else
#endif
#pragma unroll
        for (int u = 0; u < U; ++u) xj[u] = x[cj[u]];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = threadIdx.x + u * 256;
            if (i < len) s[i] = vj[u] * xj[u];
        }
        int ra[NPRE], re[NPRE];
#pragma unroll
        for (int j = 0; j < NPRE; ++j) { ra[j] = (int)(pa[j] - k0); re[j] = (int)(pe[j] - k0); }
        __syncthreads();
        auto row = [&](int r, int a, int e) {
            double v = 0.0;
            for (int i = a + sub; i < e; i += LN) v += s[i];
#pragma unroll
            for (int o = LN / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, LN);
            if (sub == 0) {
                if constexpr (DOT) { y[r0 + r] = v; acc += x[r0 + r] * v; }
                else y[r0 + r] = beta == 0.0 ? alpha * v : alpha * v + beta * y[r0 + r];
            }
        };
#pragma unroll
        for (int j = 0; j < NPRE; ++j) { const int r = g + NG * j; if (r < nr) row(r, ra[j], re[j]); }
        for (int r = g + NG * NPRE; r < nr; r += NG) row(r, (int)(rowptr[r0 + r] - k0), (int)(rowptr[r0 + r + 1] - k0)); // blocks of short rows
        __syncthreads();
        rec = recn;
    }
    if constexpr (DOT) block_sum_slots(acc, xy); // xy: a slot group (block_sum_slots below)
}

// Row-per-lane form of the same run (default).  In the kernels above lane i takes entry i, so the 64 gathers of x in one instruction follow 2.4 rows
// through all their columns: ≈ 21 scattered 24-byte pieces, and the texture-address path spends more on them than on the coalesced entry loads
// (profiling build without the gather: 0.57 instead of 0.81 ms).  Here the entries of the run are parked in LDS as they come (values and columns,
// coalesced), and the products are taken row-wise: three lanes per row, lane (row, s) the entries s, s + 3, …, so the lanes of a wave — 21 consecutive
// rows — gather x at three stencil offsets of 21 consecutive rows: a few cache lines per instruction on FE numberings.  Row sums stay in registers
// (no product array, no second LDS pass), the three partial sums meet by two lane shifts, y is stored by the lanes s = 0.
template <int CAP, bool DOT>
__global__ void __launch_bounds__(256)
k_spmv_stream_rows(int n_blk, const uint4 *__restrict__ blkrec, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ colidx, const double *__restrict__ nz,
                   const double *__restrict__ x, double alpha, double beta, double *__restrict__ y, double *__restrict__ xy)
{
    __shared__ double s_v[CAP];
    __shared__ int32_t s_c[CAP];
    constexpr int U = CAP / 256, SUB = 3, RW = 21, RP = 4 * RW, NK = 9; // 21 rows per wave (lane 63 idle): no row triple straddles two waves
    const int tid = threadIdx.x, lane = tid & 63, rl = RW * (tid >> 6) + lane / SUB, sub = lane % SUB;
    const bool lane_ok = lane < SUB * RW;
    double acc = 0.0;
    int b = blockIdx.x;
    uint4 rec = blkrec[b < n_blk ? b : 0];
    for (; b < n_blk; b += gridDim.x) {
        const int bn = b + gridDim.x;
        const uint4 recn = blkrec[bn < n_blk ? bn : b];
        const int r0 = (int)rec.x, nr = (int)(rec.y & 0xffffu), len = (int)(rec.y >> 16);
        const int64_t k0 = (int64_t)(((uint64_t)rec.w << 32) | rec.z);
        const int rc0 = rl < nr ? rl : nr - 1;
        const int64_t pa0 = rowptr[r0 + rc0], pe0 = rowptr[r0 + rc0 + 1];
        const double *nzb = nz + k0;
        const int32_t *cb = colidx + k0;
        int32_t cj[U];
        double vj[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = tid + u * 256, ic = i < len ? i : len - 1;
            cj[u] = 0; vj[u] = 0.0;
            if (len > 0) { cj[u] = cb[ic]; vj[u] = nzb[ic]; } // wave-uniform condition
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = tid + u * 256;
            if (i < len) { s_c[i] = cj[u]; s_v[i] = vj[u]; }
        }
        __syncthreads();
        for (int p0 = 0; p0 < nr; p0 += RP) {
            const int r = rl + p0;
            const bool active = lane_ok && r < nr && len > 0;
            int64_t pa = pa0, pe = pe0;
            if (p0 > 0) { const int rc = r < nr ? r : nr - 1; pa = rowptr[r0 + rc]; pe = rowptr[r0 + rc + 1]; } // runs of short rows
            const int a = (int)(pa - k0), e = (int)(pe - k0);
            int kc[NK], cc[NK];
            double vv[NK], xx[NK];
#pragma unroll
            for (int t_ = 0; t_ < NK; ++t_) {
                const int k = a + sub + SUB * t_;
                kc[t_] = active && k < e ? k : -1;
                const int kk = kc[t_] >= 0 ? kc[t_] : 0;
                cc[t_] = s_c[kk]; vv[t_] = s_v[kk];
            }
#pragma unroll
            for (int t_ = 0; t_ < NK; ++t_) xx[t_] = len > 0 ? x[cc[t_]] : 0.0;
            double v = 0.0;
#pragma unroll
            for (int t_ = 0; t_ < NK; ++t_) v += kc[t_] >= 0 ? vv[t_] * xx[t_] : 0.0;
            if (active) for (int k = a + sub + SUB * NK; k < e; k += SUB) v += s_v[k] * x[s_c[k]]; // rows longer than 27 entries
            v += __shfl_down(v, 1, 64) + __shfl_down(v, 2, 64);
            if (active && sub == 0) {
                if constexpr (DOT) { y[r0 + r] = v; acc += x[r0 + r] * v; }
                else y[r0 + r] = beta == 0.0 ? alpha * v : alpha * v + beta * y[r0 + r];
            }
        }
        __syncthreads();
        rec = recn;
    }
    if constexpr (DOT) block_sum_slots(acc, xy); // xy: a slot group (block_sum_slots below)
}

// Index-compressed form of the row-per-lane run (default when the pattern compresses).  On a finite-element numbering almost every row holds the same
// column offsets relative to its own index — the 27-point stencil of a hexahedral mesh: one signature covers 97 % of the rows at 216³, the boundary
// layers of the first-visit numbering add ≈ 10⁵ more — so the 4 B column index per non-zero is redundant: a row carries the position of its
// signature in a table (4 B per row; the table is a few MB and stays in L2), the kernel streams 8 B per non-zero instead of 12 and parks values
// only in LDS.  Lane mapping, order of the products and of the partial sums are those of k_spmv_stream_rows: the two kernels give identical bits.
// Same interface (tb_spmv_csr: the plan is built with the pattern's first product); patterns that do not compress keep the CSR kernel.
template <int CAP, bool DOT>
__global__ void __launch_bounds__(256)
k_spmv_sig_rows(int n_blk, const uint4 *__restrict__ blkrec, const int64_t *__restrict__ rowptr, const uint32_t *__restrict__ rowsig, const int32_t *__restrict__ sigoff,
                const double *__restrict__ nz, int64_t nnz, const double *__restrict__ x, double alpha, double beta, double *__restrict__ y, double *__restrict__ xy)
{
    __shared__ double2 s_v2[CAP / 2 + 1];
    double *s_v = (double *)s_v2;
    constexpr int U = CAP / 512, SUB = 3, RW = 21, RP = 4 * RW, NK = 9; // 21 rows per wave (lane 63 idle): no row triple straddles two waves
    const int tid = threadIdx.x, lane = tid & 63, rl = RW * (tid >> 6) + lane / SUB, sub = lane % SUB;
    const bool lane_ok = lane < SUB * RW;
    const int G = gridDim.x;
    double acc = 0.0;
    // column offsets of the wave's current signature, lane (row, sub) holding entries sub, sub + 3, …: re-read from the table only when a pass meets
    // another signature (97 % of the rows of a hexahedral mesh carry the interior stencil, so almost never)
    int of[NK];
    uint32_t cur = 0xFFFFFFFFu;
#pragma unroll
    for (int t_ = 0; t_ < NK; ++t_) of[t_] = 0;
    // A run costs two dependent trips to memory — its values (+ the row offsets and signatures of its first pass), then the gather of x — and the
    // kernel is bound by them, not by bytes (8 µs per run and workgroup at 216³ with 12 B or 8 B per non-zero alike).  The values of the NEXT run
    // are therefore requested while the current one is multiplied: records two runs ahead, values one run ahead (16 bytes per lane from the
    // 16-byte boundary at or below the run's first entry — 8-byte loads stream at ≈ 0.6 of that rate —, the run then sits in LDS shifted by
    // o = k0 & 1, its entry e at s_v[e + o]; the launcher guarantees a 16-byte aligned nz and runs of at most CAP − 2 entries; the one pair that
    // would reach past the array (odd nnz) is read as a single value).
    auto request = [&](const uint4 &rc, bool live, double2(&v)[U], int64_t &pa, int64_t &pe, uint32_t &sg) {
        const int r0 = (int)rc.x, nr = (int)(rc.y & 0xffffu), len = (int)(rc.y >> 16);
        const int64_t k0 = (int64_t)(((uint64_t)rc.w << 32) | rc.z);
        const int rc0 = rl < nr ? rl : nr - 1;
        pa = pe = 0; sg = 0;
        if (live) { pa = rowptr[r0 + rc0]; pe = rowptr[r0 + rc0 + 1]; sg = rowsig[r0 + rc0]; }
        const int64_t ka = k0 - (k0 & 1);
        const int npairs = (len + (int)(k0 & 1) + 1) >> 1;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = tid + u * 256;
            v[u] = make_double2(0.0, 0.0);
            if (live && i < npairs) {
                if (ka + 2 * (int64_t)i + 1 < nnz) v[u] = *(const double2 *)(nz + ka + 2 * (int64_t)i);
                else v[u].x = nz[ka + 2 * (int64_t)i];
            }
        }
    };
    int b = blockIdx.x;
    uint4 rec = blkrec[b < n_blk ? b : 0];
    uint4 recn = blkrec[b + G < n_blk ? b + G : 0];
    double2 vj[U];
    int64_t pa0, pe0;
    uint32_t sg0;
    request(rec, b < n_blk, vj, pa0, pe0, sg0);
    for (; b < n_blk; b += G) {
        const uint4 recnn = blkrec[b + 2 * G < n_blk ? b + 2 * G : 0];
        const int r0 = (int)rec.x, nr = (int)(rec.y & 0xffffu), len = (int)(rec.y >> 16);
        const int64_t k0 = (int64_t)(((uint64_t)rec.w << 32) | rec.z);
        const int o = (int)(k0 & 1);
        const int npairs = (len + o + 1) >> 1;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = tid + u * 256;
            if (i < npairs) s_v2[i] = vj[u];
        }
        __syncthreads();
        double2 vjn[U];
        int64_t pa0n = 0, pe0n = 0;
        uint32_t sg0n = 0;
        for (int p0 = 0; p0 < nr; p0 += RP) {
            const int r = rl + p0;
            const bool active = lane_ok && r < nr && len > 0;
            int64_t pa = pa0, pe = pe0;
            uint32_t sg = sg0;
            const int rc = r < nr ? r : nr - 1;
            if (p0 > 0) { pa = rowptr[r0 + rc]; pe = rowptr[r0 + rc + 1]; sg = rowsig[r0 + rc]; } // runs of short rows
            const int a = (int)(pa - k0) + o, n = (int)(pe - pa), row = r0 + rc;
            {
                const uint32_t sg1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)sg);
                const bool uniform = __ballot(sg != sg1) == 0ull; // every lane carries a valid row's signature (rc is clamped)
                if (!uniform || sg1 != cur) {
#pragma unroll
                    for (int t_ = 0; t_ < NK; ++t_) { const int k = sub + SUB * t_; of[t_] = sigoff[sg + (k < n ? k : 0)]; }
                    cur = uniform ? sg1 : 0xFFFFFFFFu;
                }
            }
            int kc[NK], cc[NK];
            double vv[NK], xx[NK];
#pragma unroll
            for (int t_ = 0; t_ < NK; ++t_) {
                const int k = sub + SUB * t_;
                kc[t_] = active && k < n ? k : -1;
                const int kk = kc[t_] >= 0 ? kc[t_] : 0;
                cc[t_] = row + (kc[t_] >= 0 ? of[t_] : 0); // masked entries read x[row]
                vv[t_] = s_v[a + kk];
            }
#pragma unroll
            for (int t_ = 0; t_ < NK; ++t_) xx[t_] = len > 0 ? x[cc[t_]] : 0.0;
            if (p0 == 0) request(recn, b + G < n_blk, vjn, pa0n, pe0n, sg0n); // behind the gather in program order: the wait for x does not include these
            double v = 0.0;
#pragma unroll
            for (int t_ = 0; t_ < NK; ++t_) v += kc[t_] >= 0 ? vv[t_] * xx[t_] : 0.0;
            if (active) for (int k = sub + SUB * NK; k < n; k += SUB) v += s_v[a + k] * x[row + sigoff[sg + k]]; // rows longer than 27 entries
            v += __shfl_down(v, 1, 64) + __shfl_down(v, 2, 64);
            if (active && sub == 0) {
                if constexpr (DOT) { y[r0 + r] = v; acc += x[r0 + r] * v; }
                else y[r0 + r] = beta == 0.0 ? alpha * v : alpha * v + beta * y[r0 + r];
            }
        }
        if (nr == 0) request(recn, b + G < n_blk, vjn, pa0n, pe0n, sg0n);
        __syncthreads();
        rec = recn; recn = recnn;
#pragma unroll
        for (int u = 0; u < U; ++u) vj[u] = vjn[u];
        pa0 = pa0n; pe0 = pe0n; sg0 = sg0n;
    }
    if constexpr (DOT) block_sum_slots(acc, xy); // xy: a slot group (block_sum_slots below)
}

// Wave-private form of k_spmv_sig_rows (TB_SPMV_KERNEL=wave; measured: 0.668 against 0.682 ms at 216³, 0.112 against 0.098 ms on the 27-layer slab — not the
// default; with the values requested TWO runs ahead it took 0.857 ms: gfx9 retires vector-memory operations in order, so the wait for a run's gather of x then
// includes the value loads issued just before it).  A run is what ONE wave multiplies — at most 21 rows and WCAP entries — and a wave
// walks its runs on its own: values of the next run requested while the current one is multiplied, its LDS slice written and read by the same wave, so the
// kernel has no workgroup barrier at all (the block form waits twice per run of ≈ 75 rows for its slowest wave).  Same lane mapping inside the wave, same
// order of products and sums: identical bits.
constexpr int SPMV_WCAP = 640;
template <bool DOT>
__global__ void __launch_bounds__(256)
k_spmv_sig_wave(int n_run, const uint4 *__restrict__ runrec, const int64_t *__restrict__ rowptr, const uint32_t *__restrict__ rowsig, const int32_t *__restrict__ sigoff,
                const double *__restrict__ nz, int64_t nnz, const double *__restrict__ x, double alpha, double beta, double *__restrict__ y, double *__restrict__ xy)
{
    constexpr int U = SPMV_WCAP / 128, SUB = 3, RW = 21, NK = 9;
    __shared__ double2 s_all[4][SPMV_WCAP / 2 + 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, rl = lane / SUB, sub = lane % SUB;
    double2 *s_v2 = s_all[wv];
    const double *s_v = (const double *)s_v2;
    const bool lane_ok = lane < SUB * RW;
    const int G = gridDim.x * 4;
    double acc = 0.0;
    int of[NK];
    uint32_t cur = 0xFFFFFFFFu;
#pragma unroll
    for (int t_ = 0; t_ < NK; ++t_) of[t_] = 0;
    auto request = [&](const uint4 &rc, bool live, double2(&v)[U], int64_t &pa, int64_t &pe, uint32_t &sg) {
        const int r0 = (int)rc.x, nr = (int)(rc.y & 0xffffu), len = (int)(rc.y >> 16);
        const int64_t k0 = (int64_t)(((uint64_t)rc.w << 32) | rc.z);
        const int rc0 = rl < nr ? rl : nr - 1;
        pa = pe = 0; sg = 0;
        if (live && nr > 0) { pa = rowptr[r0 + rc0]; pe = rowptr[r0 + rc0 + 1]; sg = rowsig[r0 + rc0]; }
        const int64_t ka = k0 - (k0 & 1);
        const int npairs = (len + (int)(k0 & 1) + 1) >> 1;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = lane + u * 64;
            v[u] = make_double2(0.0, 0.0);
            if (live && i < npairs) {
                if (ka + 2 * (int64_t)i + 1 < nnz) v[u] = *(const double2 *)(nz + ka + 2 * (int64_t)i);
                else v[u].x = nz[ka + 2 * (int64_t)i];
            }
        }
    };
    int b = blockIdx.x * 4 + wv;
    uint4 rec = runrec[b < n_run ? b : 0];
    uint4 recn = runrec[b + G < n_run ? b + G : 0];
    double2 vj[U];
    int64_t pa0, pe0;
    uint32_t sg0;
    request(rec, b < n_run, vj, pa0, pe0, sg0);
    for (; b < n_run; b += G) {
        const uint4 recnn = runrec[b + 2 * G < n_run ? b + 2 * G : 0];
        const int r0 = (int)rec.x, nr = (int)(rec.y & 0xffffu), len = (int)(rec.y >> 16);
        const int64_t k0 = (int64_t)(((uint64_t)rec.w << 32) | rec.z);
        const int o = (int)(k0 & 1);
        const int npairs = (len + o + 1) >> 1;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = lane + u * 64;
            if (i < npairs) s_v2[i] = vj[u];
        }
        __builtin_amdgcn_wave_barrier(); // LDS operations of one wave execute in order; this only keeps the compiler from moving the reads up
        double2 vjn[U];
        int64_t pa0n = 0, pe0n = 0;
        uint32_t sg0n = 0;
        {
            const int r = rl;
            const bool active = lane_ok && r < nr && len > 0;
            const int rc = r < nr ? r : (nr > 0 ? nr - 1 : 0);
            const int a = (int)(pa0 - k0) + o, n = (int)(pe0 - pa0), row = r0 + rc;
            {
                const uint32_t sg1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)sg0);
                const bool uniform = __ballot(sg0 != sg1) == 0ull;
                if (nr > 0 && (!uniform || sg1 != cur)) {
#pragma unroll
                    for (int t_ = 0; t_ < NK; ++t_) { const int k = sub + SUB * t_; of[t_] = sigoff[sg0 + (k < n ? k : 0)]; }
                    cur = uniform ? sg1 : 0xFFFFFFFFu;
                }
            }
            int kc[NK], cc[NK];
            double vv[NK], xx[NK];
#pragma unroll
            for (int t_ = 0; t_ < NK; ++t_) {
                const int k = sub + SUB * t_;
                kc[t_] = active && k < n ? k : -1;
                const int kk = kc[t_] >= 0 ? kc[t_] : 0;
                cc[t_] = row + (kc[t_] >= 0 ? of[t_] : 0);
                vv[t_] = len > 0 ? s_v[a + kk] : 0.0;
            }
#pragma unroll
            for (int t_ = 0; t_ < NK; ++t_) xx[t_] = len > 0 ? x[cc[t_]] : 0.0;
            request(recn, b + G < n_run, vjn, pa0n, pe0n, sg0n); // behind the gather in program order
            double v = 0.0;
#pragma unroll
            for (int t_ = 0; t_ < NK; ++t_) v += kc[t_] >= 0 ? vv[t_] * xx[t_] : 0.0;
            if (active) for (int k = sub + SUB * NK; k < n; k += SUB) v += s_v[a + k] * x[row + sigoff[sg0 + k]]; // rows longer than 27 entries
            v += __shfl_down(v, 1, 64) + __shfl_down(v, 2, 64);
            if (active && sub == 0) {
                if constexpr (DOT) { y[r0 + r] = v; acc += x[r0 + r] * v; }
                else y[r0 + r] = beta == 0.0 ? alpha * v : alpha * v + beta * y[r0 + r];
            }
        }
        __builtin_amdgcn_wave_barrier();
        rec = recn; recn = recnn;
#pragma unroll
        for (int u = 0; u < U; ++u) vj[u] = vjn[u];
        pa0 = pa0n; pe0 = pe0n; sg0 = sg0n;
    }
    if constexpr (DOT) block_sum_slots(acc, xy); // xy: a slot group (block_sum_slots below)
}

__global__ void __launch_bounds__(256)
k_absmax(int64_t n, const double *__restrict__ x, int64_t stride_x, unsigned long long *__restrict__ out)
{
    double m = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) m = fmax(m, fabs(x[i * stride_x]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
    __shared__ double sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3]));
        // non-negative doubles order like their bit patterns
        atomicMax(out, (unsigned long long)__double_as_longlong(m));
    }
}

// order-preserving map double → uint64 (total order of finite values and ±∞), so atomicMax works on signed values
__device__ __forceinline__ unsigned long long ordered_key(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

__global__ void __launch_bounds__(256)
k_max(int64_t n, const double *__restrict__ x, int64_t stride_x, unsigned long long *__restrict__ out)
{
    double m = -__builtin_huge_val();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) m = fmax(m, x[i * stride_x]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, ordered_key(m));
}

int launch_heat_matrix(tb_device *dev, int64_t nnz, const double *M, const double *K, double dt, double *A)
{
    hipLaunchKernelGGL(k_heat_matrix, dim3(grid_for(dev, (nnz + 1) / 2, 256)), dim3(256), 0, dev->stream, nnz, M, K, dt, A);
    TB_HIP(hipGetLastError());
    return TB_OK;
}

int launch_axpy(tb_device *dev, int64_t n, double a, const double *x, double *y)
{
    hipLaunchKernelGGL(k_axpy, dim3(grid_for(dev, n, 256)), dim3(256), 0, dev->stream, n, a, x, y);
    TB_HIP(hipGetLastError());
    return TB_OK;
}

// Vector fields (3 dofs per node, interleaved): rows 3R, 3R+1, 3R+2 share one set of columns and the columns come in triples, so the matrix
// is a CSR of 3×3 blocks stored row by row.  The block SpMV reads one column index per block (4 B per 9 values instead of 36 B) and gathers
// each x triple once for the three rows: 8.4 B per non-zero instead of 12.  G lanes per node row, each lane one block per pass.
template <int G, bool DOT>
__global__ void __launch_bounds__(256)
k_spmv_b3(int64_t n_brows, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ bcol, const double *__restrict__ nz,
          const double *__restrict__ x, double alpha, double beta, double *__restrict__ y, double *__restrict__ xy)
{
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int sub = threadIdx.x % G;
    const int64_t ngroups = ((int64_t)gridDim.x * blockDim.x) / G;
    double acc = 0.0;
    for (int64_t R = gid / G; R < n_brows; R += ngroups) {
        const int64_t k0 = rowptr[3 * R], k1 = rowptr[3 * R + 1], k2 = rowptr[3 * R + 2];
        const int nb = (int)((k1 - k0) / 3);
        const int32_t *bc = bcol + k0 / 9; // the three rows of every node row above have equal lengths: block offset = k0 / 9
        double v0 = 0.0, v1 = 0.0, v2 = 0.0;
        for (int j = sub; j < nb; j += G) {
            const int64_t c = 3 * (int64_t)bc[j];
            const double x0 = x[c], x1 = x[c + 1], x2 = x[c + 2];
            const double *a0 = nz + k0 + 3 * j, *a1 = nz + k1 + 3 * j, *a2 = nz + k2 + 3 * j;
            v0 += a0[0] * x0 + a0[1] * x1 + a0[2] * x2;
            v1 += a1[0] * x0 + a1[1] * x1 + a1[2] * x2;
            v2 += a2[0] * x0 + a2[1] * x1 + a2[2] * x2;
        }
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) {
            v0 += __shfl_xor(v0, o, G);
            v1 += __shfl_xor(v1, o, G);
            v2 += __shfl_xor(v2, o, G);
        }
        if (sub == 0) {
            double *yr = y + 3 * R;
            if constexpr (DOT) {
                yr[0] = v0; yr[1] = v1; yr[2] = v2;
                acc += x[3 * R] * v0 + x[3 * R + 1] * v1 + x[3 * R + 2] * v2;
            } else if (beta == 0.0) {
                yr[0] = alpha * v0; yr[1] = alpha * v1; yr[2] = alpha * v2;
            } else {
                yr[0] = alpha * v0 + beta * yr[0]; yr[1] = alpha * v1 + beta * yr[1]; yr[2] = alpha * v2 + beta * yr[2];
            }
        }
    }
    if constexpr (DOT) block_sum_slots(acc, xy); // xy: a slot group (block_sum_slots below)
}

// is the pattern a CSR of 3×3 blocks?  (checked once on the host; b3 = 1 yes / −1 no)
static int block3_plan(tb_pattern *p)
{
    if (p->b3 != 0) return TB_OK;
    TB_NO_CAPTURE(p->mesh->dev); // a plan is built (host work + blocking uploads) at its first use: make that use before the capture
    static const bool off = tune_env("TB_SPMV_B3") && atoi(tune_env("TB_SPMV_B3")) == 0;
    p->b3 = -1;
    if (off || p->n_rows % 3 != 0 || p->nnz % 9 != 0 || p->nnz == 0) return TB_OK;
    std::vector<int32_t> bcol((size_t)(p->nnz / 9));
    const int64_t nbr = p->n_rows / 3;
    for (int64_t R = 0; R < nbr; ++R) {
        const int64_t k0 = p->h_rowptr[3 * R], k1 = p->h_rowptr[3 * R + 1], k2 = p->h_rowptr[3 * R + 2], k3 = p->h_rowptr[3 * R + 3];
        const int64_t L = k1 - k0;
        if (L % 3 != 0 || k2 - k1 != L || k3 - k2 != L || k0 % 9 != 0) return TB_OK;
        for (int64_t j = 0; j < L; j += 3) {
            const int32_t c = p->h_colidx[k0 + j];
            if (c % 3 != 0 || p->h_colidx[k0 + j + 1] != c + 1 || p->h_colidx[k0 + j + 2] != c + 2) return TB_OK;
        }
        for (int64_t j = 0; j < L; ++j)
            if (p->h_colidx[k1 + j] != p->h_colidx[k0 + j] || p->h_colidx[k2 + j] != p->h_colidx[k0 + j]) return TB_OK;
        for (int64_t j = 0; j < L; j += 3) bcol[(size_t)(k0 / 9 + j / 3)] = p->h_colidx[k0 + j] / 3;
    }
    TB_HIP(hipMalloc((void **)&p->d_bcol, bcol.size() * sizeof(int32_t)));
    TB_HIP(hipMemcpy(p->d_bcol, bcol.data(), bcol.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    const double avg = (double)bcol.size() / (double)nbr; // blocks per node row: 27 for Q1, 64…125 for Q2
    static const int lanes_env = tune_env("TB_SPMV_B3_LANES") ? atoi(tune_env("TB_SPMV_B3_LANES")) : 0;
    p->b3_lanes = lanes_env ? lanes_env : (avg > 36 ? 32 : 16); // measured: 16 ≈ 32 > 64 on Q2 (24³ contraction solve 4.7 / 4.7 / 5.2 s), 16 best on Q1
    p->b3 = 1;
    return TB_OK;
}

template <bool DOT>
static void launch_b3(tb_pattern *p, const double *nz, const double *x, double alpha, double beta, double *y, double *xy)
{
    tb_device *dev = p->mesh->dev;
    const int64_t nbr = p->n_rows / 3;
#define TB_B3(G) hipLaunchKernelGGL((k_spmv_b3<G, DOT>), dim3(grid_for(dev, nbr * G, 256)), dim3(256), 0, dev->stream, nbr, p->d_rowptr, p->d_bcol, nz, x, alpha, beta, y, xy)
    if (p->b3_lanes == 64) TB_B3(64);
    else if (p->b3_lanes == 32) TB_B3(32);
    else TB_B3(16);
#undef TB_B3
}

// row runs of the stream SpMV: greedy cuts of the row sequence at ≤ SPMV_CAP non-zeros; n_blk = −1 (lanes-per-row kernel instead) if a single
// row exceeds the capacity
#ifndef TB_SPMV_CAP
#define TB_SPMV_CAP 2048
#endif
constexpr int SPMV_CAP = TB_SPMV_CAP;
static int stream_plan(tb_pattern *p)
{
    PlanTimer timer("stream_plan");
    if (p->n_blk != 0) return TB_OK;
    TB_NO_CAPTURE(p->mesh->dev);
    std::vector<int32_t> cut{0};
    int64_t start = 0;
    for (int64_t r = 0; r < p->n_rows; ++r) {
        if (p->h_rowptr[r + 1] - p->h_rowptr[r] > SPMV_CAP - 2) { p->n_blk = -1; return TB_OK; }
        // − 2: the 16-byte loads of the compressed kernel start one entry early and end one late (empty rows: the record holds 16 bits of row count)
        if (p->h_rowptr[r + 1] - p->h_rowptr[start] > SPMV_CAP - 2 || r - start >= 60000) { cut.push_back((int32_t)r); start = r; }
    }
    cut.push_back((int32_t)p->n_rows);
    TB_HIP(hipMalloc((void **)&p->d_blkrow, cut.size() * sizeof(int32_t)));
    TB_HIP(hipMemcpy(p->d_blkrow, cut.data(), cut.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    // one record per block for k_spmv_stream_rec: first row, rows | entries << 16, first nz (low, high word)
    static_assert(SPMV_CAP < 65536, "row and entry counts of a block share one 32-bit word");
    std::vector<uint32_t> rec(4 * (cut.size() - 1));
    for (size_t b = 0; b + 1 < cut.size(); ++b) {
        const int64_t k0 = p->h_rowptr[cut[b]], len = p->h_rowptr[cut[b + 1]] - k0;
        rec[4 * b] = (uint32_t)cut[b];
        rec[4 * b + 1] = (uint32_t)(cut[b + 1] - cut[b]) | (uint32_t)len << 16;
        rec[4 * b + 2] = (uint32_t)((uint64_t)k0 & 0xffffffffu);
        rec[4 * b + 3] = (uint32_t)((uint64_t)k0 >> 32);
    }
    TB_HIP(hipMalloc((void **)&p->d_blkrec, rec.size() * sizeof(uint32_t)));
    TB_HIP(hipMemcpy(p->d_blkrec, rec.data(), rec.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    p->n_blk = (int64_t)cut.size() - 1;
    return TB_OK;
}

// Signature plan of the index-compressed SpMV: rows with the same list of column offsets (colidx[k] − row) share one table entry.  Built on the
// host with the pattern's first product: per row a 64-bit hash (parallel), de-duplication in row order with the neighbouring row as the fast path
// (consecutive rows of a finite-element numbering nearly always repeat the signature).  The pattern "compresses" when the table is at most a
// quarter of the column array and every offset list fits the kernel (row length ≤ SPMV_CAP is checked by the stream plan); otherwise n_sig = −1.
static int sig_plan(tb_pattern *p, bool forced = false)
{
    PlanTimer timer("sig_plan");
    if (p->n_sig != 0) return TB_OK;
    TB_NO_CAPTURE(p->mesh->dev);
    const bool off = !forced && spmv_kernel_env() && strcmp(spmv_kernel_env(), "sig") != 0; // "rows" / "rec" / "chain": the CSR kernels (read per pattern: A/B runs and the bit-identity test build one pattern of each kind in one process)
    const int64_t n = p->n_rows;
    if (off || n == 0 || p->nnz >= (int64_t)0xffffffffll) { p->n_sig = -1; return TB_OK; }
    const int64_t *rp = p->h_rowptr.data();
    const int32_t *ci = p->h_colidx.data();
    std::vector<uint64_t> hsh((size_t)n);
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; ++r) {
        uint64_t h = 0x9e3779b97f4a7c15ull ^ (uint64_t)(rp[r + 1] - rp[r]);
        for (int64_t k = rp[r]; k < rp[r + 1]; ++k) {
            h ^= (uint64_t)(uint32_t)(ci[k] - (int32_t)r) + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
            h *= 0xff51afd7ed558ccdull; h ^= h >> 33;
        }
        hsh[r] = h;
    }
    std::vector<uint32_t> rowsig((size_t)n);
    std::vector<int32_t> tab;
    std::unordered_map<uint64_t, std::vector<uint32_t>> seen; // hash → positions of the signatures with that hash
    const int64_t budget = std::max<int64_t>(p->nnz / 4, 64);
    auto same = [&](uint32_t at, int64_t r) {
        const int64_t len = rp[r + 1] - rp[r];
        if ((int64_t)at + len > (int64_t)tab.size()) return false;
        for (int64_t k = 0; k < len; ++k) if (tab[at + k] != ci[rp[r] + k] - (int32_t)r) return false;
        return true;
    };
    std::vector<int32_t> siglen; // length of the signature starting at a table position is implied by the row: equal hash + equal length + equal offsets
    std::unordered_map<uint32_t, int32_t> len_at;
    int64_t nsig = 0;
    for (int64_t r = 0; r < n; ++r) {
        const int64_t len = rp[r + 1] - rp[r];
        if (r > 0 && hsh[r] == hsh[r - 1] && rp[r] - rp[r - 1] == len && same(rowsig[r - 1], r)) { rowsig[r] = rowsig[r - 1]; continue; }
        auto &cand = seen[hsh[r]];
        bool found = false;
        for (uint32_t at : cand) if (len_at[at] == (int32_t)len && same(at, r)) { rowsig[r] = at; found = true; break; }
        if (found) continue;
        const uint32_t at = (uint32_t)tab.size();
        for (int64_t k = 0; k < len; ++k) tab.push_back(ci[rp[r] + k] - (int32_t)r);
        if (len == 0) tab.push_back(0); // an empty row still owns a (never read) position
        cand.push_back(at); len_at[at] = (int32_t)len; rowsig[r] = at; ++nsig;
        if ((int64_t)tab.size() > budget) { p->n_sig = -1; return TB_OK; } // an unstructured numbering: every row its own signature
    }
    tab.resize(tab.size() + 32, 0); // the kernel reads offset 0 of a row's signature for its masked entries, and whole triples: slack at the end
    TB_HIP(hipMalloc((void **)&p->d_rowsig, rowsig.size() * sizeof(uint32_t)));
    TB_HIP(hipMemcpy(p->d_rowsig, rowsig.data(), rowsig.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    TB_HIP(hipMalloc((void **)&p->d_sigoff, tab.size() * sizeof(int32_t)));
    TB_HIP(hipMemcpy(p->d_sigoff, tab.data(), tab.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    p->n_sig = nsig; p->sig_entries = (int64_t)tab.size();
    p->h_rowsig = std::move(rowsig); // (the slice table of the mirror marks the slices whose rows share one signature)
    if (getenv("TB_PLAN_VERBOSE"))
        fprintf(stderr, "[tbhip] SpMV signature plan: %lld rows, %lld signatures, table %lld entries (%.4f of the column array)\n", (long long)n, (long long)nsig,
                (long long)tab.size(), (double)tab.size() / (double)std::max<int64_t>(p->nnz, 1));
    return TB_OK;
}

static int sig_plan_forced(tb_pattern *p) { return sig_plan(p, true); }
// runs of the wave-private kernel: ≤ 21 rows and ≤ SPMV_WCAP − 2 entries each; n_wrun = −1 when a row is longer than that
static int wave_plan(tb_pattern *p)
{
    if (p->n_wrun != 0) return TB_OK;
    TB_NO_CAPTURE(p->mesh->dev);
    std::vector<uint32_t> rec;
    int64_t start = 0;
    auto push = [&](int64_t r0, int64_t r1) {
        const int64_t k0 = p->h_rowptr[r0], len = p->h_rowptr[r1] - k0;
        rec.push_back((uint32_t)r0); rec.push_back((uint32_t)(r1 - r0) | (uint32_t)len << 16);
        rec.push_back((uint32_t)((uint64_t)k0 & 0xffffffffu)); rec.push_back((uint32_t)((uint64_t)k0 >> 32));
    };
    for (int64_t r = 0; r < p->n_rows; ++r) {
        if (p->h_rowptr[r + 1] - p->h_rowptr[r] > SPMV_WCAP - 2) { p->n_wrun = -1; return TB_OK; }
        if (p->h_rowptr[r + 1] - p->h_rowptr[start] > SPMV_WCAP - 2 || r - start >= 21) { push(start, r); start = r; }
    }
    if (p->n_rows > start) push(start, p->n_rows);
    TB_HIP(hipMalloc((void **)&p->d_wrunrec, rec.size() * sizeof(uint32_t)));
    TB_HIP(hipMemcpy(p->d_wrunrec, rec.data(), rec.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    p->n_wrun = (int64_t)rec.size() / 4;
    return TB_OK;
}

// ---- sliced mirror (tb_spmv_mirror) ----------------------------------------------------------------------------------------------------------
// The Krylov solves multiply one fixed matrix many times, and the CSR order is the wrong order for that on a wide machine: a lane that owns a row
// meets its values 216 bytes apart, so every kernel above parks the run in LDS first (load → LDS → barrier → LDS → product; 0.68 ms at 216³, 3.5
// TB/s).  The mirror stores the same values slice by slice — 64 consecutive rows, entry k of all 64 rows side by side, zero-padded to the longest
// row of the slice — so the product is one coalesced 512-byte load per entry and wave, the x gather, and the sums: no LDS, no barrier (0.47 ms,
// 5 TB/s; scripts/microbench/sell_spmv.hip).  Column offsets come from the row's signature as in k_spmv_sig_rows (scalar loads when the slice
// shares one signature, which is the rule on a hexahedral mesh), and a row's partial sums are formed in that kernel's order — entries k ≡ 0, 1, 2
// (mod 3) ascending, then s₀ + (s₁ + s₂) — so the two products agree bit for bit.  The mirror is a second copy of the values (built in ≈ 1 ms
// at 216³) bound to the array it was taken from: the caller re-binds after changing the matrix (include/tbhip.h).
struct MirrorSlice { int64_t base, obase; uint32_t sig, width; uint32_t pad[2]; }; // 32 bytes
constexpr uint32_t MIRROR_MIXED = 0xFFFFFFFFu;
constexpr int32_t MIRROR_NONE = INT32_MIN; // column offset of a padding entry
static int mirror_plan(tb_pattern *p)
{
    PlanTimer timer("mirror_plan");
    if (p->n_slices != 0) return TB_OK;
    TB_NO_CAPTURE(p->mesh->dev);
    int rc = spmv_plans(p);
    if (rc) return rc;
    if (p->b3 > 0 || p->n_rows == 0) { p->n_slices = -1; return TB_OK; }
    const bool have_sig = p->n_sig > 0 && (int64_t)p->h_rowsig.size() == p->n_rows; // a numbering without shared signatures: every slice carries its offsets
    const int64_t ns = (p->n_rows + 63) / 64;
    // one record per slice: {first value, first column offset, signature shared by its 64 rows or MIXED, width}.  A slice of one signature needs no
    // per-row metadata (offsets by scalar loads from the signature table); a mixed slice — the two ends of a grid line meet in one slice out of three at
    // 216³ — carries its column offsets entry-major like the values (4 B per entry, padding marked), so both kinds cost two trips: record → values
    // and offsets → x
    std::vector<MirrorSlice> rec((size_t)ns + 1);
    std::vector<int32_t> offs;
    int64_t at = 0;
    for (int64_t s = 0; s < ns; ++s) {
        const int64_t r0 = 64 * s, r1 = std::min<int64_t>(r0 + 64, p->n_rows);
        int64_t w = 0;
        bool uni = have_sig && r1 - r0 == 64;
        for (int64_t r = r0; r < r1; ++r) {
            w = std::max<int64_t>(w, p->h_rowptr[r + 1] - p->h_rowptr[r]);
            uni = uni && p->h_rowsig[r] == p->h_rowsig[r0];
        }
        if (w > 255) { p->n_slices = -1; return TB_OK; }
        rec[s] = MirrorSlice{at, uni ? -1 : (int64_t)offs.size(), uni ? p->h_rowsig[r0] : MIRROR_MIXED, (uint32_t)w, {0, 0}};
        if (!uni) {
            const size_t o0 = offs.size();
            offs.resize(o0 + (size_t)(64 * w), MIRROR_NONE);
            for (int64_t r = r0; r < r1; ++r)
                for (int64_t k = p->h_rowptr[r]; k < p->h_rowptr[r + 1]; ++k) offs[o0 + (size_t)(64 * (k - p->h_rowptr[r]) + (r - r0))] = p->h_colidx[k] - (int32_t)r;
        }
        at += 64 * w;
    }
    rec[ns] = MirrorSlice{at, -1, MIRROR_MIXED, 0, {0, 0}};
    MirrorSlice *db = nullptr;
    TB_HIP(hipMalloc((void **)&db, rec.size() * sizeof(MirrorSlice)));
    TB_HIP(hipMemcpy(db, rec.data(), rec.size() * sizeof(MirrorSlice), hipMemcpyHostToDevice));
    if (offs.empty()) offs.push_back(0);
    TB_HIP(hipMalloc((void **)&p->d_mir_off, offs.size() * sizeof(int32_t)));
    TB_HIP(hipMemcpy(p->d_mir_off, offs.data(), offs.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    p->d_mir_base = db; p->mir_entries = at; p->n_slices = ns;
    if (getenv("TB_PLAN_VERBOSE")) {
        int64_t mixed = 0;
        for (int64_t s = 0; s < ns; ++s) mixed += rec[s].sig == MIRROR_MIXED;
        fprintf(stderr, "[tbhip] SpMV mirror plan: %lld slices (%lld of mixed signatures), %lld value slots for %lld non-zeros\n", (long long)ns, (long long)mixed, (long long)at,
                (long long)p->nnz);
    }
    return TB_OK;
}

// values of one slice, CSR → [k][lane]: the slice's entries are one contiguous range of the value array — copied to LDS coalesced, read back
// transposed (slices wider than the LDS block read their rows directly)
__global__ void __launch_bounds__(256)
k_mirror_fill(int64_t n_rows, int64_t n_slices, const MirrorSlice *__restrict__ slices, const int64_t *__restrict__ rowptr, const double *__restrict__ nz,
              double *__restrict__ mir)
{
    constexpr int CAPW = 64 * 32;
    __shared__ double s[4][CAPW];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t slice = (int64_t)blockIdx.x * 4 + wv;
    if (slice >= n_slices) return; // waves are independent (no workgroup barrier below)
    const int64_t r0 = slice * 64, r1 = r0 + 64 < n_rows ? r0 + 64 : n_rows;
    const int64_t row = r0 + lane < n_rows ? r0 + lane : n_rows - 1;
    const int64_t pa = rowptr[row], k0 = rowptr[r0];
    const int n = r0 + lane < n_rows ? (int)(rowptr[row + 1] - pa) : 0;
    const int total = (int)(rowptr[r1] - k0);
    const int64_t b0 = slices[slice].base;
    const int W = (int)slices[slice].width;
    double *dst = mir + b0 + lane;
    if (total <= CAPW) {
        for (int i0 = 0; i0 < total; i0 += 64 * 8) { // eight loads in flight per lane
            double t[8];
#pragma unroll
            for (int u_ = 0; u_ < 8; ++u_) { const int i = i0 + 64 * u_ + lane; t[u_] = i < total ? __builtin_nontemporal_load(nz + k0 + i) : 0.0; }
#pragma unroll
            for (int u_ = 0; u_ < 8; ++u_) { const int i = i0 + 64 * u_ + lane; if (i < total) s[wv][i] = t[u_]; }
        }
        __builtin_amdgcn_wave_barrier();
        const int a = (int)(pa - k0);
#pragma unroll 9
        for (int k = 0; k < W; ++k) __builtin_nontemporal_store(k < n ? s[wv][a + k] : 0.0, dst + 64 * k);
    } else {
        for (int k = 0; k < W; ++k) dst[64 * k] = k < n ? nz[pa + k] : 0.0;
    }
}

template <bool DOT>
__global__ void __launch_bounds__(256)
k_spmv_mirror(int64_t n_rows, int64_t n_slices, const MirrorSlice *__restrict__ slices, const int32_t *__restrict__ moff, const int32_t *__restrict__ sigoff,
              const double *__restrict__ mir, const double *__restrict__ x, double alpha, double beta, double *__restrict__ y, double *__restrict__ xy)
{
    constexpr int NK = 27;
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    double acc = 0.0;
    for (int64_t slice = wave0; slice < n_slices; slice += nwaves) {
        const MirrorSlice rec = slices[slice]; // wave-uniform: scalar loads
        const int W = (int)rec.width;
        const double *vs = mir + rec.base + lane;
        const int64_t row = slice * 64 + lane;
        const bool ok = row < n_rows;
        const int64_t rc = ok ? row : n_rows - 1; // (lanes past the last row: all their entries are padding)
        double vv[NK], xx[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) vv[k] = k < W ? __builtin_nontemporal_load(vs + 64 * k) : 0.0;
        bool on[NK]; // entry k belongs to the lane's row (k < its length)
        double s0 = 0.0, s1 = 0.0, s2 = 0.0;
        if (rec.sig != MIRROR_MIXED) { // 64 rows of one signature: offsets by scalar loads, every row as wide as the slice
#pragma unroll
            for (int k = 0; k < NK; ++k) { on[k] = k < W; xx[k] = on[k] ? x[row + sigoff[rec.sig + k]] : 0.0; }
        } else {
            const int32_t *os = moff + rec.obase + lane;
            int32_t oo[NK];
#pragma unroll
            for (int k = 0; k < NK; ++k) oo[k] = k < W ? __builtin_nontemporal_load(os + 64 * k) : MIRROR_NONE;
#pragma unroll
            for (int k = 0; k < NK; ++k) { on[k] = oo[k] != MIRROR_NONE; xx[k] = on[k] ? x[rc + oo[k]] : 0.0; }
        }
#pragma unroll
        for (int t_ = 0; t_ < NK / 3; ++t_) { // (the expressions of k_spmv_sig_rows: identical rounding)
            s0 += on[3 * t_] ? vv[3 * t_] * xx[3 * t_] : 0.0;
            s1 += on[3 * t_ + 1] ? vv[3 * t_ + 1] * xx[3 * t_ + 1] : 0.0;
            s2 += on[3 * t_ + 2] ? vv[3 * t_ + 2] * xx[3 * t_ + 2] : 0.0;
        }
        for (int k = NK; k < W; k += 3) { // rows longer than 27 entries
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (k + j >= W) break;
                const int32_t o = rec.sig != MIRROR_MIXED ? sigoff[rec.sig + k + j] : moff[rec.obase + 64 * (k + j) + lane];
                if (o != MIRROR_NONE) {
                    const double a = vs[64 * (k + j)], b = x[rc + o];
                    if (j == 0) s0 += a * b; else if (j == 1) s1 += a * b; else s2 += a * b;
                }
            }
        }
        const double v = s0 + (s1 + s2);
        if (ok) {
            if constexpr (DOT) { y[row] = v; acc += x[row] * v; }
            else y[row] = beta == 0.0 ? alpha * v : alpha * v + beta * y[row];
        }
    }
    if constexpr (DOT) block_sum_slots(acc, xy); // xy: a slot group (block_sum_slots below)
}

int launch_mirror_bind(tb_pattern *p, const double *nz)
{
    tb_device *dev = p->mesh->dev;
    if (!nz) { for (const double *&q : p->mir_nz) q = nullptr; return TB_OK; }
    int rc = mirror_plan(p);
    if (rc) return rc;
    if (p->n_slices <= 0) { set_error("tb_spmv_mirror: this pattern has no sliced mirror (3x3-block rows, or rows longer than 255 entries)"); return TB_ERR_UNSUPPORTED; }
    // the slot already bound to this array (a refresh), else a free one, else the one bound longest ago
    int slot = -1;
    for (int i = 0; i < tb_pattern::MIRRORS; ++i) if (p->mir_nz[i] == nz) slot = i;
    if (slot < 0) for (int i = 0; i < tb_pattern::MIRRORS; ++i) if (!p->mir_nz[i]) { slot = i; break; }
    if (slot < 0) { slot = 0; for (int i = 1; i < tb_pattern::MIRRORS; ++i) if (p->mir_stamp[i] < p->mir_stamp[slot]) slot = i; } // least recently bound OR refreshed
    p->mir_stamp[slot] = ++p->mir_clock;
    if (!p->d_mir[slot]) {
        const size_t bytes = (size_t)p->mir_entries * sizeof(double);
        hipError_t e = hipMalloc((void **)&p->d_mir[slot], bytes);
        if (e != hipSuccess) { set_error("tb_spmv_mirror: value mirror (%zu B): %s", bytes, hipGetErrorString(e)); return TB_ERR_NOMEM; }
    }
    hipLaunchKernelGGL(k_mirror_fill, dim3((unsigned)((p->n_slices + 3) / 4)), dim3(256), 0, dev->stream, p->n_rows, p->n_slices, (const MirrorSlice *)p->d_mir_base, p->d_rowptr, nz,
                       p->d_mir[slot]);
    TB_HIP(hipGetLastError());
    p->mir_nz[slot] = nz;
    return TB_OK;
}

template <bool DOT>
static void launch_mirror(tb_pattern *p, const double *mir, const double *x, double alpha, double beta, double *y, double *xy)
{
    // one slice per wave (measured at 216³: 0.52 ms against 0.57 ms with resident workgroups only).  The fused xᵀAx form was capped at 48 workgroups per CU
    // while every workgroup ended in an atomic on ONE scalar (all 40 000: 0.62 ms); with the partials in reduction slots the cap costs 2–4 % and is gone
    const int64_t grid_env = tune_env("TB_SPMV_MIRROR_GRID") ? atoll(tune_env("TB_SPMV_MIRROR_GRID")) : 0; // (read per launch: sweeps)
    const int64_t cap = grid_env > 0 ? grid_env : (int64_t)1 << 30;
    const unsigned grid = (unsigned)std::min<int64_t>((p->n_slices + 3) / 4, cap);
    hipLaunchKernelGGL((k_spmv_mirror<DOT>), dim3(grid), dim3(256), 0, p->mesh->dev->stream, p->n_rows, p->n_slices, (const MirrorSlice *)p->d_mir_base, p->d_mir_off, p->d_sigoff,
                       mir, x, alpha, beta, y, xy);
}

// TB_SPMV_KERNEL=chain: the five-trip kernel, kept as the comparison build
static bool spmv_chain_kernel()
{
    static const bool chain = spmv_kernel_env() && !strcmp(spmv_kernel_env(), "chain");
    return chain;
}
template <bool DOT>
static void launch_stream(tb_pattern *p, const double *nz, const double *x, double alpha, double beta, double *y, double *xy, unsigned grid)
{
    hipStream_t st = p->mesh->dev->stream;
#ifdef TB_ABLATION
    static bool once = false;
    if (!once) { once = true; const int v = getenv("TB_SPMV_NOGATHER") ? 1 : 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_spmv_nogather), &v, sizeof(int)); }
#endif
    for (int i = 0; i < tb_pattern::MIRRORS; ++i) // the caller bound a sliced mirror of this very array
        if (p->mir_nz[i] == nz && nz) { launch_mirror<DOT>(p, p->d_mir[i], x, alpha, beta, y, xy); return; }
    static const bool rows_kernel = !(spmv_kernel_env() && strcmp(spmv_kernel_env(), "rows") != 0 && strcmp(spmv_kernel_env(), "sig") != 0 &&
                                      strcmp(spmv_kernel_env(), "wave") != 0); // "rec" / "chain": entry-per-lane kernels
    const bool wave_kernel = spmv_kernel_env() && !strcmp(spmv_kernel_env(), "wave"); // read per launch: the bit-identity test switches it inside one process
    if (wave_kernel && ((uintptr_t)nz & 15) == 0 && sig_plan_forced(p) == TB_OK && p->n_sig > 0 && wave_plan(p) == TB_OK && p->n_wrun > 0) {
        static int per_cu_w = 0;
        if (!per_cu_w) {
            if (tune_env("TB_SPMV_WG_PER_CU")) per_cu_w = atoi(tune_env("TB_SPMV_WG_PER_CU"));
            if (per_cu_w <= 0 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_w, (const void *)k_spmv_sig_wave<DOT>, 256, 0) != hipSuccess || per_cu_w <= 0)) per_cu_w = 3;
        }
        const unsigned gmaxw = (unsigned)(p->mesh->dev->n_cu * per_cu_w);
        hipLaunchKernelGGL((k_spmv_sig_wave<DOT>), dim3(std::min<unsigned>((unsigned)((p->n_wrun + 3) / 4), gmaxw)), dim3(256), 0, st, (int)p->n_wrun, (const uint4 *)p->d_wrunrec,
                           p->d_rowptr, p->d_rowsig, p->d_sigoff, nz, (int64_t)p->nnz, x, alpha, beta, y, xy);
        return;
    }
    if (rows_kernel && ((uintptr_t)nz & 15) == 0 && sig_plan(p) == TB_OK && p->n_sig > 0) { // default where the pattern compresses: 16 KB of LDS per workgroup
        // persistent: exactly the workgroups that are resident together (the runs are dealt round-robin, every workgroup gets the same share ± 1)
        static int per_cu = 0;
        if (!per_cu) {
            if (tune_env("TB_SPMV_WG_PER_CU")) per_cu = atoi(tune_env("TB_SPMV_WG_PER_CU"));
            if (per_cu <= 0 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_spmv_sig_rows<SPMV_CAP, DOT>, 256, 0) != hipSuccess || per_cu <= 0)) per_cu = 3;
        }
        const unsigned gmax = (unsigned)(p->mesh->dev->n_cu * per_cu);
        hipLaunchKernelGGL((k_spmv_sig_rows<SPMV_CAP, DOT>), dim3(std::min<unsigned>((unsigned)p->n_blk, gmax)), dim3(256), 0, st, (int)p->n_blk, (const uint4 *)p->d_blkrec, p->d_rowptr,
                           p->d_rowsig, p->d_sigoff, nz, (int64_t)p->nnz, x, alpha, beta, y, xy);
    } else if (rows_kernel) // 24 KB of LDS per workgroup: six resident per CU
        hipLaunchKernelGGL((k_spmv_stream_rows<SPMV_CAP, DOT>), dim3(grid > 1536 ? 1536 : grid), dim3(256), 0, st, (int)p->n_blk, (const uint4 *)p->d_blkrec, p->d_rowptr,
                           p->d_colidx, nz, x, alpha, beta, y, xy);
    else if (spmv_chain_kernel())
        hipLaunchKernelGGL((k_spmv_stream<SPMV_CAP, DOT>), dim3(grid), dim3(256), 0, st, (int)p->n_blk, p->d_blkrow, p->d_rowptr, p->d_colidx, nz, x, alpha, beta, y, xy);
    else
        hipLaunchKernelGGL((k_spmv_stream_rec<SPMV_CAP, DOT>), dim3(grid), dim3(256), 0, st, (int)p->n_blk, (const uint4 *)p->d_blkrec, p->d_rowptr, p->d_colidx, nz, x,
                           alpha, beta, y, xy);
}

static unsigned stream_grid(const tb_pattern *p)
{
    static const int64_t cap = tune_env("TB_SPMV_GRID") ? atoi(tune_env("TB_SPMV_GRID")) : 2048; // 256 CUs × 8 resident workgroups
    return (unsigned)std::min<int64_t>(p->n_blk, cap);
}

int spmv_plans(tb_pattern *p)
{
    int rc = block3_plan(p);
    if (rc || p->b3 > 0) return rc;
    rc = stream_plan(p);
    if (rc || p->n_blk <= 0) { if (p->n_sig == 0) p->n_sig = -1; return rc; }
    return sig_plan(p);
}

int launch_spmv(tb_pattern *p, const double *nz, const double *x, double alpha, double beta, double *y)
{
    tb_device *dev = p->mesh->dev;
    static const int lanes = tune_env("TB_SPMV_LANES") ? atoi(tune_env("TB_SPMV_LANES")) : 0;
    if (lanes == 0 && block3_plan(p) == TB_OK && p->b3 > 0) {
        launch_b3<false>(p, nz, x, alpha, beta, y, nullptr);
        TB_HIP(hipGetLastError());
        return TB_OK;
    }
    if (lanes == 0 && stream_plan(p) == TB_OK && p->n_blk > 0) {
        launch_stream<false>(p, nz, x, alpha, beta, y, nullptr, stream_grid(p));
        TB_HIP(hipGetLastError());
        return TB_OK;
    }
#define TB_SPMV(LN) hipLaunchKernelGGL(k_spmv<LN>, dim3(grid_for(dev, p->n_rows * LN, 256)), dim3(256), 0, dev->stream, p->n_rows, p->d_rowptr, p->d_colidx, nz, x, alpha, beta, y)
    switch (lanes) {
    case 2: TB_SPMV(2); break;
    case 4: TB_SPMV(4); break;
    case 8: TB_SPMV(8); break;
    case 32: TB_SPMV(32); break;
    default: TB_SPMV(16);
    }
#undef TB_SPMV
    TB_HIP(hipGetLastError());
    return TB_OK;
}

int launch_absmax(tb_device *dev, int64_t n, const double *x, int64_t stride, double *result)
{
    TB_NO_CAPTURE(dev); // the result goes to the host
    unsigned long long *d_out = (unsigned long long *)&dev->d_status->cell; // 8-byte scratch inside the status block
    TB_HIP(hipMemsetAsync(d_out, 0, sizeof(unsigned long long), dev->stream));
    hipLaunchKernelGGL(k_absmax, dim3(grid_for(dev, n, 256)), dim3(256), 0, dev->stream, n, x, stride, d_out);
    TB_HIP(hipGetLastError());
    unsigned long long bits = 0;
    TB_HIP(hipMemcpyAsync(&bits, d_out, sizeof bits, hipMemcpyDeviceToHost, dev->stream));
    TB_SYNC_STREAM(dev);
    memcpy(result, &bits, sizeof bits);
    return TB_OK;
}

double decode_ordered_key(unsigned long long k)
{
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    double v;
    memcpy(&v, &b, sizeof v);
    return v;
}

int launch_max(tb_device *dev, int64_t n, const double *x, int64_t stride, double *result)
{
    TB_NO_CAPTURE(dev); // the result goes to the host
    unsigned long long *d_out = (unsigned long long *)&dev->d_status->cell; // 8-byte scratch inside the status block
    TB_HIP(hipMemsetAsync(d_out, 0, sizeof(unsigned long long), dev->stream)); // key 0 < key(−∞)
    if (n > 0) {
        hipLaunchKernelGGL(k_max, dim3(grid_for(dev, n, 256)), dim3(256), 0, dev->stream, n, x, stride, d_out);
        TB_HIP(hipGetLastError());
    }
    unsigned long long bits = 0;
    TB_HIP(hipMemcpyAsync(&bits, d_out, sizeof bits, hipMemcpyDeviceToHost, dev->stream));
    TB_SYNC_STREAM(dev);
    *result = bits ? decode_ordered_key(bits) : -__builtin_huge_val();
    return TB_OK;
}

} // namespace tb

// ------------------------------------------------------------------------------------------------
// Preconditioned conjugate gradients for the heat step  (M − Δt K) uₙ = M uₙ₋₁ + f
// (src/solver/time/euler.jl:94-100; the tutorials use KrylovJL_CG(atol = 1e-6, rtol = 1e-5),
// docs/src/literate-tutorials/ep01_spiral-wave.jl:126-128).  Adjacent component: the
// Krylov method itself is third party (Krylov.jl) in the reference, so this is a plain textbook PCG
// with a Jacobi preconditioner; stopping test ‖r‖₂ ≤ atol + rtol·‖r₀‖₂ like Krylov.jl's cg.
// ------------------------------------------------------------------------------------------------
namespace tb {

// D⁻¹ for the Jacobi preconditioner: the position of each row's diagonal entry is a property of the pattern, found once on the host
// (scanning the rows on the device, one thread per row, cost 1.7 ms per solve at 216³ — more than a CG iteration); −1 = no diagonal stored
template <bool INVERT>
__global__ void __launch_bounds__(256)
k_extract_diag(int64_t nrows, const int64_t *__restrict__ diagpos, const double *__restrict__ nz, double *__restrict__ dinv)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    const int64_t k = diagpos[r];
    if constexpr (INVERT) dinv[r] = 1.0 / (k >= 0 ? nz[k] : 1.0);
    else dinv[r] = k >= 0 ? nz[k] : 0.0;
}

static int launch_extract_diag(tb_pattern *p, const double *nz, double *dinv, bool invert = true)
{
    tb_device *dev = p->mesh->dev;
    if (!p->d_diagpos) {
        TB_NO_CAPTURE(dev);
        std::vector<int64_t> pos((size_t)p->n_rows, -1);
        for (int64_t r = 0; r < p->n_rows; ++r)
            for (int64_t k = p->h_rowptr[r]; k < p->h_rowptr[r + 1]; ++k)
                if (p->h_colidx[k] == r) { pos[r] = k; break; }
        TB_HIP(hipMalloc((void **)&p->d_diagpos, pos.size() * sizeof(int64_t)));
        TB_HIP(hipMemcpy(p->d_diagpos, pos.data(), pos.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    }
    if (p->n_rows == 0) return TB_OK;
    if (invert) hipLaunchKernelGGL(k_extract_diag<true>, dim3((unsigned)((p->n_rows + 255) / 256)), dim3(256), 0, dev->stream, p->n_rows, p->d_diagpos, nz, dinv);
    else hipLaunchKernelGGL(k_extract_diag<false>, dim3((unsigned)((p->n_rows + 255) / 256)), dim3(256), 0, dev->stream, p->n_rows, p->d_diagpos, nz, dinv);
    TB_HIP(hipGetLastError());
    return TB_OK;
}
int launch_extract_diagonal(tb_pattern *p, const double *nz, double *diag) { return launch_extract_diag(p, nz, diag, false); }

// *out += the sum of v over the workgroup (4 … 16 waves).  ONE atomic per workgroup on one address: 12.2 ns each, serialised ("reduction slots" below) —
// the kernels that end this way and run once per solver iteration are launched as 1 024-thread workgroups, two per CU (grid_red): a quarter of the
// atomics of 256-thread workgroups at the same number of threads in flight.
__device__ __forceinline__ void block_sum_to(double v, double *out)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __shared__ double sm[16];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = sm[0] + sm[1] + sm[2] + sm[3];
        for (int k = 4; k < (int)(blockDim.x >> 6); ++k) t += sm[k];
        unsafeAtomicAdd(out, t);
    }
}

// two sums of a 256-thread block with one barrier: out[0] += Σ a, out[1] += Σ c (the two atomics leave from different waves)
__device__ __forceinline__ void block_sum2_to(double a, double c, double *out)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); c += __shfl_xor(c, o, 64); }
    __shared__ double sm2[8];
    if ((threadIdx.x & 63) == 0) { sm2[threadIdx.x >> 6] = a; sm2[4 + (threadIdx.x >> 6)] = c; }
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(out, sm2[0] + sm2[1] + sm2[2] + sm2[3]);
    if (threadIdx.x == 64) unsafeAtomicAdd(out + 1, sm2[4] + sm2[5] + sm2[6] + sm2[7]);
}

// ---- reduction slots ----
// A sum over the workgroups of a launch that ends in one atomic per workgroup on ONE address costs 12.2 ns per workgroup on MI355X — same-address (and
// same-128-byte-line) atomics serialise in L2, FP64 and integer alike (scripts/microbench/tail_atomics.hip: 2 048 co-resident workgroups, two scalars of
// one line: 51 µs; the CG update kernel on a 27-layer slab spent 23 of its 37 µs there).  The partial of workgroup b goes to slot b mod 64 of a GROUP of 64
// slots 128 B apart instead (3 µs for the same 2 048), and whoever needs the sum adds the 64 slots: the next kernel of a fused sequence (read_slots: the
// same xor tree in every wave, so every workgroup sees the same bits), or k_fold_slots (one wave) into a caller-owned scalar for the public one-kernel
// entries.  Groups live in tb_device::d_slots and are zero between uses; launches of one device are stream-ordered.
constexpr int RED_SLOTS = 64, RED_STRIDE = 16, RED_GROUP = RED_SLOTS * RED_STRIDE; // doubles
__device__ __forceinline__ void block_sum_slots(double v, double *group)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __shared__ double sms[4];
    if ((threadIdx.x & 63) == 0) sms[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(group + RED_STRIDE * (blockIdx.x & (RED_SLOTS - 1)), sms[0] + sms[1] + sms[2] + sms[3]);
}
__device__ __forceinline__ void block_sum2_slots(double a, double c, double *ga, double *gc)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); c += __shfl_xor(c, o, 64); }
    __shared__ double sms2[8];
    if ((threadIdx.x & 63) == 0) { sms2[threadIdx.x >> 6] = a; sms2[4 + (threadIdx.x >> 6)] = c; }
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(ga + RED_STRIDE * (blockIdx.x & (RED_SLOTS - 1)), sms2[0] + sms2[1] + sms2[2] + sms2[3]);
    if (threadIdx.x == 64) unsafeAtomicAdd(gc + RED_STRIDE * (blockIdx.x & (RED_SLOTS - 1)), sms2[4] + sms2[5] + sms2[6] + sms2[7]);
}
// the sum of a group, in every lane (call with all 64 lanes of the wave active)
__device__ __forceinline__ double read_slots(const double *group)
{
    double v = group[RED_STRIDE * (threadIdx.x & 63)];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// out[k] += sum of group k, the groups back to zero (k < ngroups ≤ 4; one wave per group)
__global__ void __launch_bounds__(256) k_fold_slots(double *__restrict__ groups, double *__restrict__ out, int ngroups)
{
    const int k = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (k >= ngroups) return;
    double *g = groups + (size_t)k * RED_GROUP;
    const double v = read_slots(g);
    g[RED_STRIDE * l] = 0.0;
    if (l == 0) out[k] += v;
}
static inline double *red_group(tb_device *dev, int k) { return dev->d_slots + (size_t)k * RED_GROUP; }
static inline void fold_slots(tb_device *dev, int first_group, double *d_out, int ngroups)
{
    hipLaunchKernelGGL(k_fold_slots, dim3(1), dim3(256), 0, dev->stream, red_group(dev, first_group), d_out, ngroups);
}

// r = b − Ax (Ax given), z = D⁻¹ r, p = z;  out[0] += r·z, out[1] += r·r
__global__ void __launch_bounds__(256)
k_cg_init(int64_t n, const double *__restrict__ b, const double *__restrict__ Ax, const double *__restrict__ dinv, double *__restrict__ r,
          double *__restrict__ p, double *__restrict__ out)
{
    double rz = 0.0, rr = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double ri = b[i] - Ax[i];
        const double zi = dinv ? dinv[i] * ri : ri;
        r[i] = ri; p[i] = zi;
        rz += ri * zi; rr += ri * ri;
    }
    block_sum_to(rz, out);
    __syncthreads();
    block_sum_to(rr, out + 1);
}

__global__ void __launch_bounds__(1024) k_dot(int64_t n, const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ out)
{
    double s = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) s += a[i] * b[i];
    block_sum_to(s, out);
}

// x += α p, r −= α Ap;  out[0] += r·(D⁻¹r), out[1] += r·r
__global__ void __launch_bounds__(1024)
k_cg_update(int64_t n, double alpha, const double *__restrict__ p, const double *__restrict__ Ap, const double *__restrict__ dinv,
            double *__restrict__ x, double *__restrict__ r, double *__restrict__ out)
{
    double rz = 0.0, rr = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        x[i] += alpha * p[i];
        const double ri = r[i] - alpha * Ap[i];
        r[i] = ri;
        rz += ri * (dinv ? dinv[i] * ri : ri);
        rr += ri * ri;
    }
    block_sum_to(rz, out);
    __syncthreads();
    block_sum_to(rr, out + 1);
}

// p = D⁻¹ r + β p
__global__ void __launch_bounds__(256)
k_cg_direction(int64_t n, double beta, const double *__restrict__ r, const double *__restrict__ dinv, double *__restrict__ p)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = (dinv ? dinv[i] * r[i] : r[i]) + beta * p[i];
}

// ---- device-resident CG scalars: the host only reads (pᵀAp, rᵀr) once per iteration to decide whether to go on ----
// scal[0..2]: r·z of the current / next / retired iteration (rotating), scal[3] = pᵀAp, scal[4] = rᵀr
template <int LANES>
__global__ void __launch_bounds__(256)
k_spmv_dot(int64_t nrows, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ colidx, const double *__restrict__ nz,
           const double *__restrict__ x, double *__restrict__ y, double *__restrict__ xy)
{
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int sub = threadIdx.x % LANES;
    const int64_t nsub = ((int64_t)gridDim.x * blockDim.x) / LANES;
    double acc = 0.0;
    for (int64_t r = gid / LANES; r < nrows; r += nsub) {
        const int64_t k0 = rowptr[r], k1 = rowptr[r + 1];
        double v = 0.0;
        for (int64_t k = k0 + sub; k < k1; k += LANES) v += nz[k] * x[colidx[k]];
#pragma unroll
        for (int o = LANES / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, LANES);
        if (sub == 0) { y[r] = v; acc += x[r] * v; }
    }
    block_sum_slots(acc, xy); // xy: a slot group
}

__global__ void __launch_bounds__(256)
k_cg_update_dev(int64_t n, const double *__restrict__ rz, const double *__restrict__ pAp, const double *__restrict__ p, const double *__restrict__ Ap,
                const double *__restrict__ dinv, double *__restrict__ x, double *__restrict__ r, double *__restrict__ rz_next, double *__restrict__ rr,
                double *__restrict__ flag)
{
    // pᵀAp ≤ 0 with a non-zero residual: the matrix is not positive definite — remembered in a sticky flag the host reads at its next
    // convergence check; at exact convergence (r = 0 ⇒ p = 0) the step is simply empty
    // rz, pAp, rz_next, rr: slot groups (see "reduction slots")
    const double pap = read_slots(pAp), rzv = read_slots(rz);
    const double alpha = pap > 0.0 ? rzv / pap : 0.0;
    if (!(pap > 0.0) && rzv != 0.0 && blockIdx.x == 0 && threadIdx.x == 0) *flag = pap == 0.0 ? -1e-300 : pap;
    double a = 0.0, c = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        x[i] += alpha * p[i];
        const double ri = r[i] - alpha * Ap[i];
        r[i] = ri;
        a += ri * (dinv ? dinv[i] * ri : ri);
        c += ri * ri;
    }
    block_sum2_slots(a, c, rz_next, rr);
}

// p = D⁻¹ r + (rz_next / rz) p; one thread retires the scalars the next iteration accumulates into
// (rz, rz_next, retired, pAp, rr: slot groups; rr_out: the scalar the host reads — ‖r‖² of this iteration)
__global__ void __launch_bounds__(256)
k_cg_direction_dev(int64_t n, const double *__restrict__ rz, const double *__restrict__ rz_next, double *__restrict__ retired, double *__restrict__ pAp,
                   double *__restrict__ rr, double *__restrict__ rr_out, const double *__restrict__ r, const double *__restrict__ dinv, double *__restrict__ p)
{
    const double rzv = read_slots(rz), rzn = read_slots(rz_next);
    const double beta = rzv > 0.0 ? rzn / rzv : 0.0;
    if (blockIdx.x == 0 && threadIdx.x < 64) { // nobody else touches these three groups during this launch
        const double v = read_slots(rr);
        const int l = RED_STRIDE * threadIdx.x;
        retired[l] = 0.0; pAp[l] = 0.0; rr[l] = 0.0;
        if (threadIdx.x == 0) *rr_out = v;
    }
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = (dinv ? dinv[i] * r[i] : r[i]) + beta * p[i];
}

int launch_cg(tb_pattern *pat, const double *A, const double *b, double *x, double rtol, double atol, int maxiter, int jacobi, int *iters,
              double *resnorm, bool b_is_residual)
{
    tb_device *dev = pat->mesh->dev;
    TB_NO_CAPTURE(dev); // reads scalars back (convergence looks)
    const int64_t n = pat->n_rows;
    if (!pat->d_cg_ws) TB_HIP(hipMalloc((void **)&pat->d_cg_ws, sizeof(double) * (4 * n + 8)));
    double *r = pat->d_cg_ws, *p = r + n, *Ap = p + n, *dinv = Ap + n, *scal = dinv + n;
    const unsigned g = grid_for(dev, n, 256);
    constexpr int LANES = 16; // 16 lanes per row measured best for 27-entry rows (0.96 vs 1.03 ms at 216³ with 8)
    const unsigned gs = grid_for(dev, n * LANES, 256);
    // TB_JACOBI_REUSE keeps D⁻¹ only when the slot holds the diagonal of THIS nz array (another operator of the pattern, a mass projection or a
    // Newton solve may have run in between): validity is tied to the array, the caller vouches that its values are unchanged
    if (jacobi == 1 || (jacobi == 2 && pat->cg_dinv_of != A)) { const int rcd = launch_extract_diag(pat, A, dinv); if (rcd) return rcd; }
    if (jacobi) pat->cg_dinv_of = A;
    const double *dp = jacobi ? dinv : nullptr;
    if (b_is_residual) TB_HIP(hipMemsetAsync(Ap, 0, sizeof(double) * n, dev->stream)); // r₀ = b given: nothing to subtract
    else { const int rc = launch_spmv(pat, A, x, 1.0, 0.0, Ap); if (rc) return rc; }
    if (b_is_residual && block3_plan(pat) == TB_OK && pat->b3 <= 0) stream_plan(pat); // the loop below picks its SpMV kernel from these plans
    TB_HIP(hipMemsetAsync(scal, 0, 6 * sizeof(double), dev->stream)); // scal[5]: sticky "pᵀAp ≤ 0" flag
    // k_cg_init writes r·z to out[0] and r·r to out[1]: point it at (scal[0], scal[1]) and move r·r to its slot afterwards
    hipLaunchKernelGGL(k_cg_init, dim3(g), dim3(256), 0, dev->stream, n, b, Ap, dp, r, p, scal);
    double h[3];
    TB_HIP(hipMemcpyAsync(h, scal, 2 * sizeof(double), hipMemcpyDeviceToHost, dev->stream));
    TB_SYNC_STREAM(dev);
    // the loop's sums live in slot groups of the device ("reduction slots"): pᵀAp, r·z of the current / next / retired iteration (rotating), rᵀr
    double *const g_pap = red_group(dev, 3), *const g_rr = red_group(dev, 7);
    auto g_rz = [&](int k) { return red_group(dev, 4 + k); };
    TB_HIP(hipMemsetAsync(g_pap, 0, 5 * RED_GROUP * sizeof(double), dev->stream));
    TB_HIP(hipMemcpyAsync(g_rz(0), scal, sizeof(double), hipMemcpyDeviceToDevice, dev->stream)); // r·z of the start: slot 0 of the current group
    double rnorm = std::sqrt(h[1]);
    const double tol = atol + rtol * rnorm;
    pat->last_tol = tol;
    // The host looks at (‖r‖², flag) once per `check` iterations: small systems are bound by the host round trip, not by the kernels, so they
    // run a few iterations between looks (at most check − 1 iterations past the tolerance); large ones look every iteration.
    static const int check_env = tune_env("TB_CG_CHECK_EVERY") ? atoi(tune_env("TB_CG_CHECK_EVERY")) : 0;
    const int check0 = check_env > 0 ? check_env : (n >= 262144 ? 1 : 4);
    int it = 0, cur = 0, last_look = 0;
    while (rnorm > tol && it < maxiter) {
        // long solves (elasticity: thousands of iterations) look less often still — every 4th iteration after 32, every 8th after 128 —
        // so a solve overshoots its tolerance by at most 6 % of its length, while short ones (the heat step: ~5) are checked every time
        const int check = check_env > 0 ? check_env : (it >= 128 ? 8 : it >= 32 ? std::max(check0, 4) : check0);
        const int nxt = (cur + 1) % 3, ret = (cur + 2) % 3;
        if (pat->b3 > 0)
            launch_b3<true>(pat, A, p, 1.0, 0.0, Ap, g_pap);
        else if (pat->n_blk > 0)
            launch_stream<true>(pat, A, p, 1.0, 0.0, Ap, g_pap, stream_grid(pat));
        else
            hipLaunchKernelGGL(k_spmv_dot<LANES>, dim3(gs), dim3(256), 0, dev->stream, n, pat->d_rowptr, pat->d_colidx, A, p, Ap, g_pap);
        hipLaunchKernelGGL(k_cg_update_dev, dim3(g), dim3(256), 0, dev->stream, n, g_rz(cur), g_pap, p, Ap, dp, x, r, g_rz(nxt), g_rr, scal + 5);
        const bool look = it + 1 - last_look >= check || it + 1 == maxiter;
        if (look) last_look = it + 1;
        hipLaunchKernelGGL(k_cg_direction_dev, dim3(g), dim3(256), 0, dev->stream, n, g_rz(cur), g_rz(nxt), g_rz(ret), g_pap, g_rr, scal + 4, r, dp, p);
        if (look) TB_HIP(hipMemcpyAsync(h, scal + 3, 3 * sizeof(double), hipMemcpyDeviceToHost, dev->stream)); // (‖r‖² lands in scal[4] in the direction kernel)
        cur = nxt;
        ++it;
        if (!look) continue;
        TB_SYNC_STREAM(dev);
        if (h[2] != 0.0) { set_error("tb_cg_solve: matrix is not positive definite (pᵀAp = %g)", h[2] == -1e-300 ? 0.0 : h[2]); return TB_ERR_BAD_ARG; }
        rnorm = std::sqrt(h[1]);
    }
    TB_HIP(hipGetLastError());
    if (iters) *iters = it;
    if (resnorm) *resnorm = rnorm;
    return TB_OK;
}

// ------------------------------------------------------------------------------------------------
// Restarted GMRES — the default inner solver of the reference's Newton–Raphson (LinearSolve.KrylovJL_GMRES(),
// src/solver/nonlinear/newton_raphson.jl:61; Krylov.jl is third party).  Needed where the tangent is not positive definite
// (non-polyconvex energies, follower loads).  Right Jacobi preconditioning (A D⁻¹ y = b, x = D⁻¹ y: the monitored residual is the
// true one), classical Gram–Schmidt with one re-orthogonalisation pass so that a whole Arnoldi step is eight launches and one
// host read: h = Vᵀw and w −= V h are single kernels over all basis vectors.
// ------------------------------------------------------------------------------------------------
// out[j · GM_HS] += V[j]·w for j < k (blockIdx.y = j)
constexpr int GM_HS = 16; // doubles between two Arnoldi coefficients on the device
__global__ void __launch_bounds__(256)
k_multi_dot(int64_t n, const double *__restrict__ V, const double *__restrict__ w, double *__restrict__ out)
{
    const double *v = V + (int64_t)blockIdx.y * n;
    double s = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) s += v[i] * w[i];
    block_sum_to(s, out + (size_t)blockIdx.y * GM_HS); // one 128-byte line per result: atomics on one line serialise ("reduction slots" below)
}
// w += sign · Σ_{j<k} c[j] V[j];  optionally ww += w·w of the result
__global__ void __launch_bounds__(1024)
k_multi_axpy(int64_t n, int k, double sign, const double *__restrict__ c, int cstride, const double *__restrict__ V, double *__restrict__ w, double *__restrict__ ww)
{
    double acc = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double s = 0.0;
        for (int j = 0; j < k; ++j) s += c[(size_t)j * cstride] * V[(int64_t)j * n + i];
        const double r = w[i] + sign * s;
        w[i] = r;
        acc += r * r;
    }
    if (ww) block_sum_to(acc, ww);
}
// y = a · (d ? d .* x : x)
__global__ void __launch_bounds__(256)
k_scale_diag(int64_t n, double a, const double *__restrict__ d, const double *__restrict__ x, double *__restrict__ y)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = a * (d ? d[i] * x[i] : x[i]);
}
// r = b − Ax; rr += r·r
__global__ void __launch_bounds__(256)
k_residual(int64_t n, const double *__restrict__ b, const double *__restrict__ Ax, double *__restrict__ r, double *__restrict__ rr)
{
    double acc = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) { const double v = b[i] - Ax[i]; r[i] = v; acc += v * v; }
    block_sum_to(acc, rr);
}
// x += d ? d .* t : t
__global__ void __launch_bounds__(256)
k_add_diag(int64_t n, const double *__restrict__ d, const double *__restrict__ t, double *__restrict__ x)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) x[i] += d ? d[i] * t[i] : t[i];
}

int launch_gmres(tb_pattern *pat, const double *A, const double *b, double *x, double rtol, double atol, int maxiter, int restart, int jacobi,
                 int *iters, double *resnorm)
{
    tb_device *dev = pat->mesh->dev;
    TB_NO_CAPTURE(dev); // reads scalars back (convergence looks)
    const int64_t n = pat->n_rows;
    const int m = restart;
    if (pat->gmres_m < m) {
        if (pat->d_gmres_ws) TB_HIP(hipFree(pat->d_gmres_ws));
        pat->d_gmres_ws = nullptr;
        TB_HIP(hipMalloc((void **)&pat->d_gmres_ws, sizeof(double) * ((size_t)(m + 4) * n + (2 * GM_HS + 1) * (size_t)(m + 2))));
        pat->gmres_m = m;
    }
    double *V = pat->d_gmres_ws, *w = V + (size_t)(m + 1) * n, *z = w + n, *dinv = z + n, *sc = dinv + n; // sc: h1[(m+2)·GM_HS] | h2[(m+2)·GM_HS] | y[m+2]
    double *h1 = sc, *h2 = sc + (size_t)(m + 2) * GM_HS, *yd = sc + 2 * (size_t)(m + 2) * GM_HS;
    const unsigned g = grid_for(dev, n, 256);
    if (jacobi) { const int rcd = launch_extract_diag(pat, A, dinv); if (rcd) return rcd; }
    const double *dp = jacobi ? dinv : nullptr;
    std::vector<double> H((size_t)(m + 1) * m), cs(m), sn(m), gvec(m + 1), yh(m), hh(2 * (size_t)(m + 2) * GM_HS);
    int it = 0;
    double rnorm = 0.0, tol = 0.0;
    bool first = true;
    while (true) {
        // r = b − A x → V[0] = r/‖r‖
        int rc = launch_spmv(pat, A, x, 1.0, 0.0, z);
        if (rc) return rc;
        TB_HIP(hipMemsetAsync(h1, 0, sizeof(double), dev->stream));
        hipLaunchKernelGGL(k_residual, dim3(g), dim3(256), 0, dev->stream, n, b, z, w, h1);
        TB_HIP(hipMemcpyAsync(hh.data(), h1, sizeof(double), hipMemcpyDeviceToHost, dev->stream));
        TB_SYNC_STREAM(dev);
        rnorm = std::sqrt(hh[0]);
        if (first) { tol = atol + rtol * rnorm; pat->last_tol = tol; first = false; }
        if (!(rnorm > tol) || it >= maxiter) break;
        hipLaunchKernelGGL(k_scale_diag, dim3(g), dim3(256), 0, dev->stream, n, 1.0 / rnorm, (const double *)nullptr, w, V);
        std::fill(gvec.begin(), gvec.end(), 0.0);
        gvec[0] = rnorm;
        int j = 0;
        double res_est = rnorm;
        for (; j < m && it < maxiter && res_est > tol; ++j, ++it) {
            // w = A D⁻¹ v_j
            hipLaunchKernelGGL(k_scale_diag, dim3(g), dim3(256), 0, dev->stream, n, 1.0, dp, V + (size_t)j * n, z);
            rc = launch_spmv(pat, A, z, 1.0, 0.0, w);
            if (rc) return rc;
            TB_HIP(hipMemsetAsync(sc, 0, sizeof(double) * 2 * (m + 2) * GM_HS, dev->stream));
            // every workgroup of a dot ends in one atomic on its vector's result: ≈ 8 per CU in all (j + 1 vectors share them), each result on a line of its own
            const unsigned gd = std::max(1u, std::min(g, (unsigned)dev->n_cu * 8u / (unsigned)(j + 1)));
            hipLaunchKernelGGL(k_multi_dot, dim3(gd, j + 1), dim3(256), 0, dev->stream, n, V, w, h1);
            hipLaunchKernelGGL(k_multi_axpy, dim3(grid_red(dev, n)), dim3(1024), 0, dev->stream, n, j + 1, -1.0, h1, GM_HS, V, w, (double *)nullptr);
            hipLaunchKernelGGL(k_multi_dot, dim3(gd, j + 1), dim3(256), 0, dev->stream, n, V, w, h2);
            hipLaunchKernelGGL(k_multi_axpy, dim3(grid_red(dev, n)), dim3(1024), 0, dev->stream, n, j + 1, -1.0, h2, GM_HS, V, w, h2 + (size_t)(m + 1) * GM_HS); // ‖w‖² in the last slot
            TB_HIP(hipMemcpyAsync(hh.data(), sc, sizeof(double) * 2 * (m + 2) * GM_HS, hipMemcpyDeviceToHost, dev->stream));
            TB_SYNC_STREAM(dev);
            double *Hj = H.data() + (size_t)j * (m + 1);
            for (int i = 0; i <= j; ++i) Hj[i] = hh[(size_t)i * GM_HS] + hh[((size_t)(m + 2) + i) * GM_HS];
            const double wn = std::sqrt(hh[((size_t)(m + 2) + (m + 1)) * GM_HS]);
            Hj[j + 1] = wn;
            if (!std::isfinite(wn)) { set_error("tb_gmres_solve: breakdown (non-finite Arnoldi vector)"); return TB_ERR_BAD_ARG; }
            for (int i = 0; i < j; ++i) { const double t = cs[i] * Hj[i] + sn[i] * Hj[i + 1]; Hj[i + 1] = -sn[i] * Hj[i] + cs[i] * Hj[i + 1]; Hj[i] = t; }
            const double den = std::hypot(Hj[j], Hj[j + 1]);
            cs[j] = den > 0 ? Hj[j] / den : 1.0; sn[j] = den > 0 ? Hj[j + 1] / den : 0.0;
            Hj[j] = den; Hj[j + 1] = 0.0;
            gvec[j + 1] = -sn[j] * gvec[j]; gvec[j] = cs[j] * gvec[j];
            res_est = std::fabs(gvec[j + 1]);
            if (wn > 0.0 && j + 1 <= m) hipLaunchKernelGGL(k_scale_diag, dim3(g), dim3(256), 0, dev->stream, n, 1.0 / wn, (const double *)nullptr, w, V + (size_t)(j + 1) * n);
            if (wn == 0.0) { ++j; ++it; break; } // lucky breakdown: the Krylov space is invariant, the solution is exact in it
        }
        // y = H⁻¹ g (upper triangular), x += D⁻¹ V y
        for (int i = j - 1; i >= 0; --i) {
            double sacc = gvec[i];
            for (int l = i + 1; l < j; ++l) sacc -= H[(size_t)l * (m + 1) + i] * yh[l];
            yh[i] = sacc / H[(size_t)i * (m + 1) + i];
        }
        TB_HIP(hipMemcpyAsync(yd, yh.data(), sizeof(double) * j, hipMemcpyHostToDevice, dev->stream));
        TB_HIP(hipMemsetAsync(z, 0, sizeof(double) * n, dev->stream));
        hipLaunchKernelGGL(k_multi_axpy, dim3(grid_red(dev, n)), dim3(1024), 0, dev->stream, n, j, 1.0, yd, 1, V, z, (double *)nullptr);
        hipLaunchKernelGGL(k_add_diag, dim3(g), dim3(256), 0, dev->stream, n, dp, z, x);
        TB_SYNC_STREAM(dev); // yh is reused by the next cycle
    }
    TB_HIP(hipGetLastError());
    if (iters) *iters = it;
    if (resnorm) *resnorm = rnorm;
    return TB_OK;
}

// ------------------------------------------------------------------------------------------------
// ℓ₁ Gauss–Seidel preconditioner (Baker, Falgout, Kolev, Yang, "Multigrid smoothers for ultraparallel computing", SIAM J. Sci. Comput.
// 33 (2011), §6) — the preconditioner the reference's documentation lists for its Krylov solves (Thunderbolt.Preconditioners.L1GSPrecBuilder
// with ForwardSweep / BackwardSweep / SymmetricSweep, docs/src/api-reference/solver.md:13-22; its source is not part of the reference
// checkout, so this restates the published algorithm: parity unpinned).  Rows are cut into partitions of `ps` consecutive rows; inside a
// partition the sweep is exact Gauss–Seidel, couplings that leave the partition are moved onto the diagonal by their ℓ₁ norm:
//   D̃_ii = a_ii + Σ_{j ∉ part(i)} |a_ij|;  forward: (D̃ + L_p) y = r;  symmetric: then (D̃ + U_p) z = D̃ y,
// i.e. M = (D̃ + L_p) D̃⁻¹ (D̃ + U_p), symmetric positive definite whenever A is.  One wavefront per partition: the rows are visited in
// order, the lanes share a row's entries; the partition's part of the iterate lives in LDS.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_l1gs_diag(int64_t n, int ps, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ colidx, const double *__restrict__ nz, double *__restrict__ dt)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int64_t lo = r / ps * ps, hi = lo + ps;
    double d = 0.0;
    for (int64_t k = rowptr[r]; k < rowptr[r + 1]; ++k) {
        const int32_t c = colidx[k];
        if (c == r) d += nz[k];
        else if (c < lo || c >= hi) d += fabs(nz[k]);
    }
    dt[r] = d;
}

template <bool SYMMETRIC>
__global__ void __launch_bounds__(256)
k_l1gs_apply(int64_t n, int ps, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ colidx, const double *__restrict__ nz,
             const double *__restrict__ dt, const double *__restrict__ r, double *__restrict__ z)
{
    extern __shared__ double s_y[]; // [waves per block][ps]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t part = (int64_t)blockIdx.x * (blockDim.x >> 6) + wv;
    const int64_t lo = part * ps;
    if (lo >= n) return;
    const int64_t hi = lo + ps < n ? lo + ps : n;
    double *y = s_y + (size_t)wv * ps;
    for (int64_t i = lo; i < hi; ++i) { // forward sweep
        double acc = 0.0;
        for (int64_t k = rowptr[i] + lane; k < rowptr[i + 1]; k += 64) {
            const int32_t c = colidx[k];
            if (c >= lo && c < i) acc += nz[k] * y[c - lo];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (lane == 0) y[i - lo] = (r[i] - acc) / dt[i];
        __builtin_amdgcn_wave_barrier();
    }
    if constexpr (SYMMETRIC) {
        for (int64_t i = hi - 1; i >= lo; --i) { // backward sweep: z_i = y_i − Σ_{j > i in the partition} a_ij z_j / D̃_ii, in place
            double acc = 0.0;
            for (int64_t k = rowptr[i] + lane; k < rowptr[i + 1]; k += 64) {
                const int32_t c = colidx[k];
                if (c > i && c < hi) acc += nz[k] * y[c - lo];
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
            if (lane == 0) y[i - lo] -= acc / dt[i];
            __builtin_amdgcn_wave_barrier();
        }
    }
    for (int64_t i = lo + lane; i < hi; i += 64) z[i] = y[i - lo];
}

int launch_l1gs_setup(tb_pattern *pat, const double *A, int ps, double *d_dtilde)
{
    tb_device *dev = pat->mesh->dev;
    const int64_t n = pat->n_rows;
    hipLaunchKernelGGL(k_l1gs_diag, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, dev->stream, n, ps, pat->d_rowptr, pat->d_colidx, A, d_dtilde);
    TB_HIP(hipGetLastError());
    return TB_OK;
}

int launch_l1gs_apply(tb_pattern *pat, const double *A, const double *d_dtilde, int ps, int symmetric, const double *r, double *z)
{
    tb_device *dev = pat->mesh->dev;
    const int64_t n = pat->n_rows, nparts = (n + ps - 1) / ps;
    const size_t lds = sizeof(double) * 4 * (size_t)ps;
    const dim3 grid((unsigned)((nparts + 3) / 4)), block(256);
    if (symmetric) hipLaunchKernelGGL(k_l1gs_apply<true>, grid, block, lds, dev->stream, n, ps, pat->d_rowptr, pat->d_colidx, A, d_dtilde, r, z);
    else hipLaunchKernelGGL(k_l1gs_apply<false>, grid, block, lds, dev->stream, n, ps, pat->d_rowptr, pat->d_colidx, A, d_dtilde, r, z);
    TB_HIP(hipGetLastError());
    return TB_OK;
}

// preconditioned CG with a general preconditioner application (two host reads per iteration): z = M⁻¹ r by ℓ₁ Gauss–Seidel
int launch_pcg_l1gs(tb_pattern *pat, const double *A, const double *b, double *x, double rtol, double atol, int maxiter, int ps, int *iters, double *resnorm)
{
    tb_device *dev = pat->mesh->dev;
    TB_NO_CAPTURE(dev); // reads scalars back (convergence looks)
    const int64_t n = pat->n_rows;
    if (!pat->d_pcg_ws) TB_HIP(hipMalloc((void **)&pat->d_pcg_ws, sizeof(double) * (5 * n + 8)));
    double *r = pat->d_pcg_ws, *z = r + n, *p = z + n, *Ap = p + n, *dtl = Ap + n, *scal = dtl + n;
    const unsigned g = grid_for(dev, n, 256);
    int rc = launch_l1gs_setup(pat, A, ps, dtl);
    if (rc) return rc;
    rc = launch_spmv(pat, A, x, 1.0, 0.0, Ap);
    if (rc) return rc;
    double h[2];
    TB_HIP(hipMemsetAsync(scal, 0, 2 * sizeof(double), dev->stream));
    hipLaunchKernelGGL(k_residual, dim3(g), dim3(256), 0, dev->stream, n, b, Ap, r, scal);
    TB_HIP(hipMemcpyAsync(h, scal, sizeof(double), hipMemcpyDeviceToHost, dev->stream));
    TB_SYNC_STREAM(dev);
    double rnorm = std::sqrt(h[0]);
    const double tol = atol + rtol * rnorm;
    pat->last_tol = tol;
    int it = 0;
    double rz = 0.0;
    while (rnorm > tol && it < maxiter) {
        rc = launch_l1gs_apply(pat, A, dtl, ps, 1, r, z);
        if (rc) return rc;
        TB_HIP(hipMemsetAsync(scal, 0, 2 * sizeof(double), dev->stream));
        hipLaunchKernelGGL(k_dot, dim3(grid_red(dev, n)), dim3(1024), 0, dev->stream, n, r, z, scal);
        TB_HIP(hipMemcpyAsync(h, scal, sizeof(double), hipMemcpyDeviceToHost, dev->stream));
        TB_SYNC_STREAM(dev);
        const double rz_new = h[0];
        if (it == 0) TB_HIP(hipMemcpyAsync(p, z, sizeof(double) * n, hipMemcpyDeviceToDevice, dev->stream));
        else hipLaunchKernelGGL(k_cg_direction, dim3(g), dim3(256), 0, dev->stream, n, rz_new / rz, z, (const double *)nullptr, p);
        rz = rz_new;
        rc = launch_spmv(pat, A, p, 1.0, 0.0, Ap);
        if (rc) return rc;
        TB_HIP(hipMemsetAsync(scal, 0, 2 * sizeof(double), dev->stream));
        hipLaunchKernelGGL(k_dot, dim3(grid_red(dev, n)), dim3(1024), 0, dev->stream, n, p, Ap, scal);
        TB_HIP(hipMemcpyAsync(h, scal, sizeof(double), hipMemcpyDeviceToHost, dev->stream));
        TB_SYNC_STREAM(dev);
        if (!(h[0] > 0.0)) { set_error("tb_pcg_solve: matrix is not positive definite (pᵀAp = %g)", h[0]); return TB_ERR_BAD_ARG; }
        const double alpha = rz / h[0];
        TB_HIP(hipMemsetAsync(scal, 0, 2 * sizeof(double), dev->stream));
        hipLaunchKernelGGL(k_cg_update, dim3(grid_red(dev, n)), dim3(1024), 0, dev->stream, n, alpha, p, Ap, (const double *)nullptr, x, r, scal);
        TB_HIP(hipMemcpyAsync(h, scal, 2 * sizeof(double), hipMemcpyDeviceToHost, dev->stream));
        TB_SYNC_STREAM(dev);
        rnorm = std::sqrt(h[1]);
        ++it;
    }
    TB_HIP(hipGetLastError());
    if (iters) *iters = it;
    if (resnorm) *resnorm = rnorm;
    return TB_OK;
}

// ------------------------------------------------------------------------------------------------
// Chebyshev polynomial preconditioner (the smoother the reference's multigrid extension uses, docs: "damped Jacobi with Chebyshev-optimal ω",
// src/solver/linear/multigrid.jl:28-33 — here as a preconditioner of its own: M⁻¹ = p_m(D⁻¹A)·D⁻¹ with the degree-m Chebyshev polynomial of
// the interval [λmax/ratio, λmax] of D⁻¹A).  A fixed symmetric positive-definite operator, so plain PCG applies; it needs SpMVs and one fused
// vector kernel per degree and no inner products — a degree-m application costs m − 1 products and removes about m of every m + 1 outer
// iterations, i.e. the same number of products as Jacobi-CG but 1/m of its dot products, host looks and vector kernels (elasticity tangents:
// ≈2 000 Jacobi-CG iterations on a 10⁵-dof Q2 block).  λmax from 24 Lanczos steps (largest Ritz value + 5 %), capped by the Gershgorin bound.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_gershgorin(int64_t n, const int64_t *__restrict__ rowptr, const double *__restrict__ nz, const double *__restrict__ dinv, double *__restrict__ out)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    double s = 0.0;
    for (int64_t k = rowptr[r]; k < rowptr[r + 1]; ++k) s += fabs(nz[k]);
    out[r] = s * fabs(dinv[r]);
}
// d = c1·d + c2·D⁻¹(r − w);  z += d        (w = A z; first step: w = NULL, d = c2·D⁻¹ r, z = d)
__global__ void __launch_bounds__(256)
k_cheb_step(int64_t n, double c1, double c2, const double *__restrict__ dinv, const double *__restrict__ r, const double *__restrict__ w, double *__restrict__ d,
            double *__restrict__ z)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (w) {
            const double di = c1 * d[i] + c2 * dinv[i] * (r[i] - w[i]);
            d[i] = di;
            z[i] += di;
        } else {
            const double di = c2 * dinv[i] * r[i];
            d[i] = di;
            z[i] = di;
        }
    }
}
// Lanczos helpers: sq = √D⁻¹, v = a positive start vector (the Gershgorin row sums), v₋₁ = 0
__global__ void __launch_bounds__(256)
k_lanczos_init(int64_t n, const double *__restrict__ dinv, const double *__restrict__ start, double *__restrict__ sq, double *__restrict__ v, double *__restrict__ vp)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        sq[i] = sqrt(fabs(dinv[i]));
        v[i] = start[i] * (1.0 + 0.37 * (double)((i * 2654435761u) & 1023) / 1024.0); // perturbed so that symmetric modes are not missed
        vp[i] = 0.0;
    }
}
__global__ void __launch_bounds__(256) k_mul_to(int64_t n, const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ y)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = a[i] * b[i];
}
// w = sq∘w − β v₋₁;  out += w·v
__global__ void __launch_bounds__(1024)
k_lanczos_a(int64_t n, const double *__restrict__ sq, double beta, const double *__restrict__ vp, const double *__restrict__ v, double *__restrict__ w, double *__restrict__ out)
{
    double a = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double wi = sq[i] * w[i] - beta * vp[i];
        w[i] = wi;
        a += wi * v[i];
    }
    block_sum_to(a, out);
}
// w −= α v;  out += w·w
__global__ void __launch_bounds__(1024)
k_lanczos_b(int64_t n, double alpha, const double *__restrict__ v, double *__restrict__ w, double *__restrict__ out)
{
    double a = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double wi = w[i] - alpha * v[i];
        w[i] = wi;
        a += wi * wi;
    }
    block_sum_to(a, out);
}
__global__ void __launch_bounds__(256) k_scale_to(int64_t n, double s, const double *__restrict__ x, double *__restrict__ y)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = s * x[i];
}
// p = z + β p with β = rz_new / rz from device scalars (β = 0 when rz = 0: first iteration)
__global__ void __launch_bounds__(256)
k_pcg_direction_dev(int64_t n, const double *__restrict__ rz, const double *__restrict__ rz_new, const double *__restrict__ z, double *__restrict__ p)
{
    const double beta = *rz > 0.0 ? *rz_new / *rz : 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = z[i] + beta * p[i];
}
// x += α p, r −= α Ap with α = rz_new / pAp from device scalars; rr += r·r; a non-positive pAp is remembered in *flag
__global__ void __launch_bounds__(1024)
k_pcg_update_dev(int64_t n, const double *__restrict__ rz_new, const double *__restrict__ pAp, const double *__restrict__ p, const double *__restrict__ Ap,
                 double *__restrict__ x, double *__restrict__ r, double *__restrict__ rr, double *__restrict__ flag)
{
    const double pap = *pAp;
    const double alpha = pap > 0.0 ? *rz_new / pap : 0.0;
    if (!(pap > 0.0) && *rz_new != 0.0 && blockIdx.x == 0 && threadIdx.x == 0) *flag = pap == 0.0 ? -1e-300 : pap;
    double c = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        x[i] += alpha * p[i];
        const double ri = r[i] - alpha * Ap[i];
        r[i] = ri;
        c += ri * ri;
    }
    block_sum_to(c, rr);
}

int launch_pcg_chebyshev(tb_pattern *pat, const double *A, const double *b, double *x, double rtol, double atol, int maxiter, int degree, int *iters, double *resnorm)
{
    tb_device *dev = pat->mesh->dev;
    TB_NO_CAPTURE(dev); // reads scalars back (convergence looks)
    const int64_t n = pat->n_rows;
    if (!pat->d_cheb_ws) TB_HIP(hipMalloc((void **)&pat->d_cheb_ws, sizeof(double) * (7 * n + 16)));
    double *r = pat->d_cheb_ws, *z = r + n, *p = z + n, *Ap = p + n, *dinv = Ap + n, *d = dinv + n, *w = d + n, *S = w + n; // S: rz | rz_new | pAp | rr | flag | power sums
    const unsigned g = grid_for(dev, n, 256);
    int rc = launch_extract_diag(pat, A, dinv);
    if (rc) return rc;
    // λmax(D⁻¹A) = λmax(D^-½ A D^-½): Gershgorin bound (safe, loose), sharpened by the largest Ritz value of 24 Lanczos steps (converges from
    // below, within a per cent after a few tens of steps) inflated by 5 % — an interval that misses the top of the spectrum would make the
    // polynomial indefinite there
    hipLaunchKernelGGL(k_gershgorin, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, dev->stream, n, pat->d_rowptr, A, dinv, w);
    double gersh = 0.0;
    rc = launch_absmax(dev, n, w, 1, &gersh);
    if (rc) return rc;
    double lmax = gersh;
    {
        constexpr int KL = 24;
        double al[KL], be[KL + 1], h[2];
        double *v = z, *vp = p, *t = Ap, *sq = d; // scratch: the solver's vectors are not in use yet
        hipLaunchKernelGGL(k_lanczos_init, dim3(g), dim3(256), 0, dev->stream, n, dinv, w, sq, v, vp);
        TB_HIP(hipMemsetAsync(S + 8, 0, 2 * sizeof(double), dev->stream));
        hipLaunchKernelGGL(k_dot, dim3(grid_red(dev, n)), dim3(1024), 0, dev->stream, n, v, v, S + 8);
        TB_HIP(hipMemcpyAsync(h, S + 8, sizeof(double), hipMemcpyDeviceToHost, dev->stream));
        TB_SYNC_STREAM(dev);
        int kdone = 0;
        if (h[0] > 0.0) {
            hipLaunchKernelGGL(k_scale_to, dim3(g), dim3(256), 0, dev->stream, n, 1.0 / std::sqrt(h[0]), v, v);
            be[0] = 0.0;
            for (int k = 0; k < KL; ++k) {
                hipLaunchKernelGGL(k_mul_to, dim3(g), dim3(256), 0, dev->stream, n, sq, v, t);            // t = D^-½ v
                rc = launch_spmv(pat, A, t, 1.0, 0.0, w);                                                    // w = A t
                if (rc) return rc;
                TB_HIP(hipMemsetAsync(S + 8, 0, 2 * sizeof(double), dev->stream));
                hipLaunchKernelGGL(k_lanczos_a, dim3(grid_red(dev, n)), dim3(1024), 0, dev->stream, n, sq, be[k], vp, v, w, S + 8); // w = D^-½ w − β v₋₁; α = w·v
                TB_HIP(hipMemcpyAsync(h, S + 8, sizeof(double), hipMemcpyDeviceToHost, dev->stream));
                TB_SYNC_STREAM(dev);
                al[k] = h[0];
                hipLaunchKernelGGL(k_lanczos_b, dim3(grid_red(dev, n)), dim3(1024), 0, dev->stream, n, al[k], v, w, S + 9);  // w −= α v; ‖w‖²
                TB_HIP(hipMemcpyAsync(h, S + 9, sizeof(double), hipMemcpyDeviceToHost, dev->stream));
                TB_SYNC_STREAM(dev);
                kdone = k + 1;
                be[k + 1] = std::sqrt(h[0]);
                if (!(be[k + 1] > 1e-12 * std::fabs(al[k]))) break;                                           // invariant subspace: the Ritz values are exact
                TB_HIP(hipMemcpyAsync(vp, v, sizeof(double) * n, hipMemcpyDeviceToDevice, dev->stream));
                hipLaunchKernelGGL(k_scale_to, dim3(g), dim3(256), 0, dev->stream, n, 1.0 / be[k + 1], w, v);
            }
        }
        if (kdone > 0) { // largest eigenvalue of the tridiagonal (α, β) by bisection on the Sturm count
            double hi = 0.0;
            for (int k = 0; k < kdone; ++k) hi = std::max(hi, std::fabs(al[k]) + std::fabs(be[k]) + (k + 1 < kdone ? std::fabs(be[k + 1]) : 0.0));
            double lo = 0.0;
            for (int itb = 0; itb < 80; ++itb) {
                const double x0 = 0.5 * (lo + hi);
                int above = 0; // eigenvalues greater than x0 = number of positive pivots of T − x0 I
                double q = al[0] - x0;
                if (q > 0) ++above;
                for (int k = 1; k < kdone; ++k) {
                    if (q == 0.0) q = 1e-300;
                    q = al[k] - x0 - be[k] * be[k] / q;
                    if (q > 0) ++above;
                }
                if (above > 0) lo = x0; else hi = x0;
            }
            lmax = std::min(gersh, 1.05 * hi);
        }
    }
    static const double ratio_env = tune_env("TB_CHEB_RATIO") ? atof(tune_env("TB_CHEB_RATIO")) : 0.0;
    const int m = degree < 1 ? 1 : degree;
    const double lmin = lmax / (ratio_env > 1.0 ? ratio_env : std::max(4.0, 1.8 * m * m));
    const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin), sigma1 = theta / delta;
    auto precondition = [&]() -> int { // z = p_m(D⁻¹A) D⁻¹ r  (Saad, Iterative Methods, Alg. 12.1, started from zero)
        double rho = 1.0 / sigma1;
        hipLaunchKernelGGL(k_cheb_step, dim3(g), dim3(256), 0, dev->stream, n, 0.0, 1.0 / theta, dinv, r, (const double *)nullptr, d, z);
        for (int k = 1; k < m; ++k) {
            int rc2 = launch_spmv(pat, A, z, 1.0, 0.0, w);
            if (rc2) return rc2;
            const double rho_new = 1.0 / (2.0 * sigma1 - rho);
            hipLaunchKernelGGL(k_cheb_step, dim3(g), dim3(256), 0, dev->stream, n, rho_new * rho, 2.0 * rho_new / delta, dinv, r, w, d, z);
            rho = rho_new;
        }
        return TB_OK;
    };
    rc = launch_spmv(pat, A, x, 1.0, 0.0, Ap);
    if (rc) return rc;
    double h[3];
    TB_HIP(hipMemsetAsync(S, 0, 8 * sizeof(double), dev->stream));
    hipLaunchKernelGGL(k_residual, dim3(g), dim3(256), 0, dev->stream, n, b, Ap, r, S + 3);
    TB_HIP(hipMemcpyAsync(h, S + 3, sizeof(double), hipMemcpyDeviceToHost, dev->stream));
    TB_SYNC_STREAM(dev);
    double rnorm = std::sqrt(h[0]);
    const double tol = atol + rtol * rnorm;
    pat->last_tol = tol;
    int it = 0;
    while (rnorm > tol && it < maxiter) {
        const int look_every = it >= 64 ? 4 : 1;
        for (int s2 = 0; s2 < look_every && it < maxiter; ++s2, ++it) {
            rc = precondition();
            if (rc) return rc;
            // rz ← rz_new of the previous iteration; rz_new = r·z
            TB_HIP(hipMemcpyAsync(S, S + 1, sizeof(double), hipMemcpyDeviceToDevice, dev->stream));
            TB_HIP(hipMemsetAsync(S + 1, 0, 3 * sizeof(double), dev->stream));
            hipLaunchKernelGGL(k_dot, dim3(grid_red(dev, n)), dim3(1024), 0, dev->stream, n, r, z, S + 1);
            hipLaunchKernelGGL(k_pcg_direction_dev, dim3(g), dim3(256), 0, dev->stream, n, S, S + 1, z, p);
            rc = launch_spmv(pat, A, p, 1.0, 0.0, Ap);
            if (rc) return rc;
            hipLaunchKernelGGL(k_dot, dim3(grid_red(dev, n)), dim3(1024), 0, dev->stream, n, p, Ap, S + 2);
            hipLaunchKernelGGL(k_pcg_update_dev, dim3(grid_red(dev, n)), dim3(1024), 0, dev->stream, n, S + 1, S + 2, p, Ap, x, r, S + 3, S + 4);
        }
        TB_HIP(hipMemcpyAsync(h, S + 3, 2 * sizeof(double), hipMemcpyDeviceToHost, dev->stream));
        TB_SYNC_STREAM(dev);
        if (h[1] != 0.0) { set_error("tb_pcg_solve: matrix or Chebyshev preconditioner is not positive definite (pᵀAp = %g)", h[1] == -1e-300 ? 0.0 : h[1]); return TB_ERR_BAD_ARG; }
        rnorm = std::sqrt(h[0]);
    }
    TB_HIP(hipGetLastError());
    if (iters) *iters = it;
    if (resnorm) *resnorm = rnorm;
    return TB_OK;
}

// apply_zero!(K, f, ch) on device CSR (Ferrite.apply_zero!; CSR method src/utils.jl:263-278, call sites
// src/solver/nonlinear/nlsolve_common.jl:12-26): rows and columns of prescribed dofs are zeroed, their diagonal entry is set
// to `diag` (Ferrite uses the mean diagonal so the conditioning survives), f is zeroed there.  8 lanes per row.
__global__ void __launch_bounds__(256)
k_apply_zero_csr(int64_t nrows, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ colidx, const uint8_t *__restrict__ flags,
                 double diag, double *__restrict__ nz, double *__restrict__ f)
{
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const int sub = threadIdx.x & 7;
    if (r >= nrows) return;
    const bool pr = flags[r];
    if (nz)
        for (int64_t k = rowptr[r] + sub; k < rowptr[r + 1]; k += 8) {
            const int32_t c = colidx[k];
            if (pr) nz[k] = c == r ? diag : 0.0;
            else if (flags[c]) nz[k] = 0.0;
        }
    if (f && pr && sub == 0) f[r] = 0.0;
}

// Σ |diag| / n (Ferrite.meandiag)
__global__ void __launch_bounds__(256)
k_sum_absdiag(int64_t nrows, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ colidx, const double *__restrict__ nz, double *__restrict__ out)
{
    double s = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += stride)
        for (int64_t k = rowptr[r]; k < rowptr[r + 1]; ++k)
            if (colidx[k] == r) { s += fabs(nz[k]); break; }
    block_sum_to(s, out);
}

static int scratch2(tb_pattern *pat, double **scal)
{
    const int64_t n = pat->n_rows;
    if (!pat->d_cg_ws) TB_HIP(hipMalloc((void **)&pat->d_cg_ws, sizeof(double) * (4 * n + 8)));
    *scal = pat->d_cg_ws + 4 * n;
    return TB_OK;
}

int launch_apply_zero(tb_pattern *pat, double *nz, double *f, const uint8_t *flags, double diag)
{
    tb_device *dev = pat->mesh->dev;
    const int64_t n = pat->n_rows;
    if (!n) return TB_OK;
    hipLaunchKernelGGL(k_apply_zero_csr, dim3((unsigned)((n * 8 + 255) / 256)), dim3(256), 0, dev->stream, n, pat->d_rowptr, pat->d_colidx, flags, diag, nz, f);
    TB_HIP(hipGetLastError());
    return TB_OK;
}

int launch_meandiag(tb_pattern *pat, const double *nz, double *result)
{
    tb_device *dev = pat->mesh->dev;
    TB_NO_CAPTURE(dev); // reads scalars back (convergence looks)
    const int64_t n = pat->n_rows;
    *result = 0.0;
    if (!n) return TB_OK;
    double *scal;
    int rc = scratch2(pat, &scal);
    if (rc) return rc;
    TB_HIP(hipMemsetAsync(scal, 0, 2 * sizeof(double), dev->stream));
    hipLaunchKernelGGL(k_sum_absdiag, dim3(grid_for(dev, n, 256)), dim3(256), 0, dev->stream, n, pat->d_rowptr, pat->d_colidx, nz, scal);
    TB_HIP(hipGetLastError());
    double h = 0.0;
    TB_HIP(hipMemcpyAsync(&h, scal, sizeof h, hipMemcpyDeviceToHost, dev->stream));
    TB_SYNC_STREAM(dev);
    *result = h / (double)n;
    return TB_OK;
}

// ---- sub-structured CG over several devices: weighted sums (a dof held by k ranks counts 1/k), every scalar in caller-owned device memory ----
__global__ void __launch_bounds__(256)
k_cgd_dot(int64_t n, const double *__restrict__ w, const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ out /* slot group */)
{
    double s = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) s += (w ? w[i] : 1.0) * a[i] * b[i];
    block_sum_slots(s, out);
}

// α = rz / pAp (device scalars, already summed over the ranks); x += α p, r −= α Ap; out[0] += Σ w r·(D⁻¹r), out[1] += Σ w r·r
// SL: pᵀAp is still in its slot group (the fused iteration: nothing folded it); the two sums always leave through slot groups g_rz, g_rr
template <bool SL>
__global__ void __launch_bounds__(256)
k_cgd_update(int64_t n, const double *__restrict__ w, const double *__restrict__ dinv, const double *__restrict__ p, const double *__restrict__ Ap,
             double *__restrict__ x, double *__restrict__ r, const double *__restrict__ rz, const double *__restrict__ pAp, double *__restrict__ out,
             double *__restrict__ g_rz, double *__restrict__ g_rr)
{
    const double pap = SL ? read_slots(pAp) : *pAp;
    const double alpha = pap > 0.0 ? *rz / pap : 0.0;
    // pᵀAp ≤ 0 while r·z ≠ 0: the operator is not positive definite (or the iteration broke down) — sticky flag in out[2], read by the host with ‖r‖²
    if (!(pap > 0.0) && *rz != 0.0 && blockIdx.x == 0 && threadIdx.x == 0) out[2] = pap == 0.0 ? -1e-300 : pap;
    double a = 0.0, c = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        x[i] += alpha * p[i];
        const double ri = r[i] - alpha * Ap[i];
        r[i] = ri;
        const double wi = w ? w[i] : 1.0;
        a += wi * ri * (dinv ? dinv[i] * ri : ri);
        c += wi * ri * ri;
    }
    block_sum2_slots(a, c, g_rz, g_rr);
}

// β = rz_new / rz (device scalars); p = D⁻¹ r + β p
template <bool SL>
__global__ void __launch_bounds__(256)
k_cgd_direction(int64_t n, const double *__restrict__ dinv, const double *__restrict__ r, double *__restrict__ p, const double *__restrict__ rz,
                const double *__restrict__ rz_new)
{
    const double rzn = SL ? read_slots(rz_new) : *rz_new;
    const double beta = *rz > 0.0 ? rzn / *rz : 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = (dinv ? dinv[i] * r[i] : r[i]) + beta * p[i];
}

// end of an iteration: rz ← rz_new, ‖r‖² parked in S[5] for the host's next look, the three accumulators (pAp, rz_new, rr) back to zero — one
// launch for what would be two copies and a fill
__global__ void k_cgd_rotate(double *__restrict__ S)
{
    if (threadIdx.x == 0) { S[0] = S[2]; S[5] = S[3]; S[1] = 0.0; S[2] = 0.0; S[3] = 0.0; }
}
// the same at the end of a fused iteration, whose sums are still in slot groups 0 (pᵀAp), 1 (r·z), 2 (rᵀr): r·z and ‖r‖² folded into S, the groups zeroed
__global__ void __launch_bounds__(64) k_cgd_rotate_slots(double *__restrict__ S, double *__restrict__ groups)
{
    const double rz = read_slots(groups + RED_GROUP), rr = read_slots(groups + 2 * RED_GROUP);
    const int l = RED_STRIDE * threadIdx.x;
    groups[l] = 0.0; groups[RED_GROUP + l] = 0.0; groups[2 * RED_GROUP + l] = 0.0;
    if (threadIdx.x == 0) { S[0] = rz; S[5] = rr; S[1] = 0.0; S[2] = 0.0; S[3] = 0.0; }
}
int launch_cgd_rotate(tb_device *dev, double *d_S)
{
    hipLaunchKernelGGL(k_cgd_rotate, dim3(1), dim3(64), 0, dev->stream, d_S);
    TB_HIP(hipGetLastError());
    return TB_OK;
}

int launch_cgd_dot(tb_device *dev, int64_t n, const double *w, const double *a, const double *b, double *d_out)
{
    if (n > 0) {
        hipLaunchKernelGGL(k_cgd_dot, dim3(grid_for(dev, n, 256)), dim3(256), 0, dev->stream, n, w, a, b, red_group(dev, 0));
        fold_slots(dev, 0, d_out, 1);
    }
    TB_HIP(hipGetLastError());
    return TB_OK;
}
int launch_cgd_update(tb_device *dev, int64_t n, const double *w, const double *dinv, const double *p, const double *Ap, double *x, double *r,
                      const double *d_rz, const double *d_pAp, double *d_out3)
{
    if (n > 0) {
        hipLaunchKernelGGL(k_cgd_update<false>, dim3(grid_for(dev, n, 256)), dim3(256), 0, dev->stream, n, w, dinv, p, Ap, x, r, d_rz, d_pAp, d_out3,
                           red_group(dev, 1), red_group(dev, 2));
        fold_slots(dev, 1, d_out3, 2); // out3[0] += r·z, out3[1] += rᵀr
    }
    TB_HIP(hipGetLastError());
    return TB_OK;
}
int launch_cgd_direction(tb_device *dev, int64_t n, const double *dinv, const double *r, double *p, const double *d_rz, const double *d_rz_new)
{
    if (n > 0) hipLaunchKernelGGL(k_cgd_direction<false>, dim3(grid_for(dev, n, 256)), dim3(256), 0, dev->stream, n, dinv, r, p, d_rz, d_rz_new);
    TB_HIP(hipGetLastError());
    return TB_OK;
}
int launch_spmv_dot_slots(tb_pattern *pat, const double *A, const double *x, double *y, double *group);
// One iteration of the device CG on a sub-domain without shared dofs, its three sums kept in slot groups 0–2 from kernel to kernel (no fold launches):
// SpMV with pᵀAp → update (reads the group) → direction (reads the group) → rotate (folds r·z and ‖r‖² into S, zeroes the groups)
int launch_cgd_iteration(tb_pattern *pat, const double *A, const double *dinv, double *x, double *r, double *p, double *Ap, double *d_S)
{
    tb_device *dev = pat->mesh->dev;
    const int64_t n = pat->n_rows;
    if (n == 0) return launch_cgd_rotate(dev, d_S);
    int rc = launch_spmv_dot_slots(pat, A, p, Ap, red_group(dev, 0));
    if (rc) return rc;
    hipLaunchKernelGGL(k_cgd_update<true>, dim3(grid_for(dev, n, 256)), dim3(256), 0, dev->stream, n, (const double *)nullptr, dinv, p, Ap, x, r, d_S,
                       red_group(dev, 0), d_S + 2, red_group(dev, 1), red_group(dev, 2));
    hipLaunchKernelGGL(k_cgd_direction<true>, dim3(grid_for(dev, n, 256)), dim3(256), 0, dev->stream, n, dinv, r, p, d_S, red_group(dev, 1));
    hipLaunchKernelGGL(k_cgd_rotate_slots, dim3(1), dim3(64), 0, dev->stream, d_S, red_group(dev, 0));
    TB_HIP(hipGetLastError());
    return TB_OK;
}

// ---- halo pack / unpack and the packed interface rows of a product (multi-GPU path) ----
__global__ void __launch_bounds__(256) k_gather_indexed(int64_t n, const double *__restrict__ vec, const int32_t *__restrict__ idx, double *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = vec[idx[i]];
}

__global__ void __launch_bounds__(256) k_scatter_add_indexed(int64_t n, const double *__restrict__ in, const int32_t *__restrict__ idx, double *__restrict__ vec)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) vec[idx[i]] += in[i]; // indices of one call are distinct
}

__global__ void __launch_bounds__(256) k_scatter_indexed(int64_t n, const double *__restrict__ in, const int32_t *__restrict__ idx, double *__restrict__ vec)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) vec[idx[i]] = in[i];
}

// out[k] = Σ_j A[rows[k], j] x[j]: 16 lanes per listed row
__global__ void __launch_bounds__(256)
k_spmv_rows(int64_t n, const int32_t *__restrict__ rows, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ colidx, const double *__restrict__ nz,
            const double *__restrict__ x, double *__restrict__ out)
{
    constexpr int LN = 16;
    const int sub = threadIdx.x % LN;
    const int64_t nsub = ((int64_t)gridDim.x * blockDim.x) / LN;
    for (int64_t k = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / LN; k < n; k += nsub) {
        const int32_t r = rows[k];
        double v = 0.0;
        for (int64_t e = rowptr[r] + sub; e < rowptr[r + 1]; e += LN) v += nz[e] * x[colidx[e]];
#pragma unroll
        for (int o = LN / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, LN);
        if (sub == 0) out[k] = v;
    }
}

int launch_gather_indexed(tb_device *dev, int64_t n, const double *vec, const int32_t *idx, double *out)
{
    if (n > 0) hipLaunchKernelGGL(k_gather_indexed, dim3(grid_for(dev, n, 256)), dim3(256), 0, dev->stream, n, vec, idx, out);
    TB_HIP(hipGetLastError());
    return TB_OK;
}
int launch_scatter_add_indexed(tb_device *dev, int64_t n, const double *in, const int32_t *idx, double *vec)
{
    if (n > 0) hipLaunchKernelGGL(k_scatter_add_indexed, dim3(grid_for(dev, n, 256)), dim3(256), 0, dev->stream, n, in, idx, vec);
    TB_HIP(hipGetLastError());
    return TB_OK;
}
int launch_scatter_indexed(tb_device *dev, int64_t n, const double *in, const int32_t *idx, double *vec)
{
    if (n > 0) hipLaunchKernelGGL(k_scatter_indexed, dim3(grid_for(dev, n, 256)), dim3(256), 0, dev->stream, n, in, idx, vec);
    TB_HIP(hipGetLastError());
    return TB_OK;
}
int launch_spmv_rows(tb_pattern *p, const double *nz, const double *x, int64_t n, const int32_t *rows, double *out)
{
    tb_device *dev = p->mesh->dev;
    if (n > 0) hipLaunchKernelGGL(k_spmv_rows, dim3(grid_for(dev, n * 16, 256)), dim3(256), 0, dev->stream, n, rows, p->d_rowptr, p->d_colidx, nz, x, out);
    TB_HIP(hipGetLastError());
    return TB_OK;
}
// y = A x and *d_dot += xᵀ y with the kernels of the single-device CG (block / stream / lanes-per-row forms)
int launch_spmv_dot_slots(tb_pattern *pat, const double *A, const double *x, double *y, double *d_dot);
// y = A x and *d_dot += xᵀy: the kernels leave the sum in slot group 0, one wave folds it into the caller's scalar
int launch_spmv_dot(tb_pattern *pat, const double *A, const double *x, double *y, double *d_dot)
{
    if (pat->n_rows == 0) return TB_OK;
    tb_device *dev = pat->mesh->dev;
    int rc = launch_spmv_dot_slots(pat, A, x, y, red_group(dev, 0));
    if (rc) return rc;
    fold_slots(dev, 0, d_dot, 1);
    TB_HIP(hipGetLastError());
    return TB_OK;
}
int launch_spmv_dot_slots(tb_pattern *pat, const double *A, const double *x, double *y, double *d_dot /* a slot group */)
{
    tb_device *dev = pat->mesh->dev;
    const int64_t n = pat->n_rows;
    if (n == 0) return TB_OK;
    static const int lanes_env = tune_env("TB_SPMV_LANES") ? atoi(tune_env("TB_SPMV_LANES")) : 0;
    if (lanes_env == 0) { int rc = block3_plan(pat); if (rc) return rc; if (pat->b3 <= 0) { rc = stream_plan(pat); if (rc) return rc; } }
    if (lanes_env == 0 && pat->b3 > 0)
        launch_b3<true>(pat, A, x, 1.0, 0.0, y, d_dot);
    else if (lanes_env == 0 && pat->n_blk > 0)
        launch_stream<true>(pat, A, x, 1.0, 0.0, y, d_dot, stream_grid(pat));
    else
        hipLaunchKernelGGL(k_spmv_dot<16>, dim3(grid_for(dev, n * 16, 256)), dim3(256), 0, dev->stream, n, pat->d_rowptr, pat->d_colidx, A, x, y, d_dot);
    TB_HIP(hipGetLastError());
    return TB_OK;
}

int launch_dot(tb_device *dev, int64_t n, const double *a, const double *b, double *result)
{
    TB_NO_CAPTURE(dev); // the result goes to the host
    double *scal = (double *)&dev->d_status->cell; // 8-byte scratch inside the status block
    TB_HIP(hipMemsetAsync(scal, 0, sizeof(double), dev->stream));
    if (n > 0) {
        hipLaunchKernelGGL(k_dot, dim3(grid_red(dev, n)), dim3(1024), 0, dev->stream, n, a, b, scal);
        TB_HIP(hipGetLastError());
    }
    TB_HIP(hipMemcpyAsync(result, scal, sizeof(double), hipMemcpyDeviceToHost, dev->stream));
    TB_SYNC_STREAM(dev);
    return TB_OK;
}

} // namespace tb
