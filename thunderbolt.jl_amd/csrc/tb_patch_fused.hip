// tb_patch_fused.hip — PATCH-strategy assembly of the mass and diffusion matrices on trilinear hexahedra (2×2×2 Gauss rule),
// either one of them or BOTH in one pass over the mesh.
//
// Reference call sites: the backward-Euler heat stage assembles the mass and the diffusion operator of one DofHandler back to
// back on the same sparsity pattern (src/solver/time/euler.jl:172-176, pattern sharing :110-116); element routines
// src/modeling/core/mass.jl:28-43, src/modeling/core/diffusion.jl:28-50, geometry src/ferrite-addons/PR883.jl:253-291,367-387.
//
// One workgroup per patch (tb_plans.cpp): a thread integrates one cell instance with the sum-factorised element routine
// (tb_hex8_sumfac.hpp: geometry once for both forms, ≈1 400 FP64 instructions per cell for Kₑ and Mₑ together), adds the
// entries of the rows its patch owns into LDS row accumulators (ds_add_f64; one block of accumulators per matrix) and every
// row leaves as one contiguous store per matrix: each nz is written exactly once, no zero-fill pass, no global atomics.
// Per-instance metadata is 20 B (8 patch-local node indices + signature index); the 64 positions "column dof(j) in row
// dof(i)" of a cell come from a de-duplicated signature table; the vertex coordinates of a patch's nodes are pre-gathered per patch
// (contiguous, coalesced) and staged in LDS.  Measured on MI355X at 216³ (profiles/r02_*): M + K in one pass 1.9 ms against
// 3.7 ms for the two launches of the general patch kernel of tb_assembly.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#include "tb_elem.hpp"
#include "tb_forms.hpp"
#include "tb_hex8_sumfac.hpp"
#include "tb_internal.h"

namespace tb {
using namespace tbk;

struct FusedView {
    const int64_t *elem_ptr, *row_ptr, *node_ptr;
    const int32_t *elem_cell;
    const uint16_t *elem_ln;
    const uint32_t *elem_sig;
    const uint8_t *sigtab;
    const int32_t *pnode;
    const RowDesc *row_desc;
    int kcap, max_rows, max_nodes;
#ifdef TB_ABLATION
    long long *prof;
#endif
};

// One cell instance: geometry of the 8 points (once for both forms), sum-factorised Kₑ / Mₑ, entries of owned rows added to the LDS row
// accumulators.  x: vertex coordinates, cp: the cell's 64 positions (byte i*8+j; dwords 2i, 2i+1 hold row i), ro[i]: entry offset of row i
// in the accumulator block or 0xFFFFFFFF when the patch does not own it.
// meta(cp, ro) delivers the scatter metadata when the scatter starts (the streaming kernel reads them from LDS only then: held from the start, they
// cost 24 registers through the contraction).
template <bool WK, bool WM, bool FK, bool FM, bool DIAG = false, bool ISO = false, class Meta>
__device__ __forceinline__ void hex8_instance(const double (&x)[8][3], Meta &&meta, int64_t cell, const FormArgs &faK,
                                              const FormArgs &faM, double *accK, double *accM, Status *st, const int32_t *elem_cell, int64_t e)
{
    double G[8][6], dq[8];
    if constexpr (FK) {
        // field tensors: mesh and coefficient field are fixed, so G_q and detJ_q were tabulated once (k_tabulate_g_hex8: the same
        // hex8_sf_geometry, detJ ≤ 0 reported there) — the cell needs no geometry stage and no coordinates, only its 56 numbers
        const double *gp = faK.gtab + cell * 56;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
#pragma unroll
            for (int k = 0; k < 6; ++k) G[q][k] = gp[7 * q + k];
            dq[q] = gp[7 * q + 6];
        }
    } else {
        auto Dq = [&](int, double(&D)[6]) { D[0] = faK.D[0]; D[1] = faK.D[1]; D[2] = faK.D[2]; D[3] = faK.D[4]; D[4] = faK.D[5]; D[5] = faK.D[8]; };
#ifdef TB_ABLATION
        if (faK.debug & 4) { // no arithmetic: keep the loads alive, feed the scatter with trivial values
#pragma unroll
            for (int q = 0; q < 8; ++q) { dq[q] = x[q][0] + x[q][1] + x[q][2]; for (int k = 0; k < 6; ++k) G[q][k] = dq[q]; }
        } else
#endif
        if (!hex8_sf_geometry<WK, false, DIAG, ISO>(x, Dq, G, dq, ISO ? faK.iso_scale : -0.25)) { st->neg_detj = 1; st->cell = elem_cell ? elem_cell[e] : -1; }
    }
    Hex8SFK c;
    if constexpr (WK) hex8_sf_contract(G, c);
    uint4 cp[4];
    uint32_t ro[8];
    meta(cp, ro);
    bool own[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) own[i] = ro[i] != 0xFFFFFFFFu;
#ifdef TB_ABLATION
    if (faK.debug & 1) { // no LDS adds: one add per thread keeps the values alive
#pragma unroll
        for (int i = 1; i < 8; ++i) own[i] = false;
        own[0] = own[0] && (G[0][0] + dq[0] == 1.2345);
    }
#endif
    auto pos = [&](int i, int j) -> uint32_t {
        const uint4 &c4 = cp[i >> 1];
        const uint32_t w = (i & 1) ? (j < 4 ? c4.z : c4.w) : (j < 4 ? c4.x : c4.y);
        return (w >> (8 * (j & 3))) & 0xffu;
    };
    if constexpr (WK) {
        // every entry goes to LDS as soon as it is formed (rows i and j of the symmetric pair): Kₑ is never held as a whole
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = i; j < 8; ++j) {
                const double v = hex8_sf_entry(c, i, j);
                if (own[i]) unsafeAtomicAdd(accK + (ro[i] + pos(i, j)), v);
                if (j > i && own[j]) unsafeAtomicAdd(accK + (ro[j] + pos(j, i)), v);
            }
    }
    if constexpr (WM) {
        if constexpr (FM) { // ρ at the Gauss points from first-order nodal data per cell (FieldCoefficient, coefficients.jl:85-99)
            double rn[8];
#pragma unroll
            for (int a = 0; a < 8; ++a) rn[a] = faM.field[cell * 8 + a];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                double r = 0.0;
#pragma unroll
                for (int a = 0; a < 8; ++a) r += Hex8<2>::N(q, a) * rn[a];
                dq[q] *= r;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) dq[q] *= faM.rho;
        }
        double Z[3][3][3];
        hex8_sf_mass(dq, Z);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (own[i]) {
#pragma unroll
                for (int j = 0; j < 8; ++j) unsafeAtomicAdd(accM + (ro[i] + pos(i, j)), Z[SF::ty(i, j, 0)][SF::ty(i, j, 1)][SF::ty(i, j, 2)]);
            }
    }
}

// G_q and detJ_q of every cell for a diffusion form with a field tensor (tabulated D_q in, 7 doubles per point out)
__global__ void __launch_bounds__(256)
k_tabulate_g_hex8(MeshView m, const double *__restrict__ dtab, double *__restrict__ gtab, Status *st)
{
    const int64_t cell = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= m.n_cells) return;
    double x[8][3];
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        const double *px = m.xyz + 3 * (int64_t)m.conn[cell * 8 + a];
        x[a][0] = px[0]; x[a][1] = px[1]; x[a][2] = px[2];
    }
    double G[8][6], dq[8];
    auto Dq = [&](int q, double(&D)[6]) {
        const double *dp = dtab + (cell * 8 + q) * 6;
#pragma unroll
        for (int k = 0; k < 6; ++k) D[k] = dp[k];
    };
    if (!hex8_sf_geometry<true, false>(x, Dq, G, dq)) { st->neg_detj = 1; st->cell = cell; }
    double *gp = gtab + cell * 56;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
#pragma unroll
        for (int k = 0; k < 6; ++k) gp[7 * q + k] = G[q][k];
        gp[7 * q + 6] = dq[q];
    }
}

// General form: one workgroup per patch, any number of instances / rows / nodes per patch.
template <bool WK, bool WM, bool FK, bool FM>
__global__ void __launch_bounds__(256, 2)
k_patch_hex8(MeshView m, FormArgs faK, FormArgs faM, FusedView pv, double *__restrict__ nzK, double *__restrict__ nzM, Status *st)
{
    extern __shared__ double lds[];
    const int T = blockDim.x, tid = threadIdx.x;
    const int64_t p = blockIdx.x;
    const int64_t e0 = pv.elem_ptr[p], e1 = pv.elem_ptr[p + 1];
    const int64_t r0 = pv.row_ptr[p];
    const int nrows = (int)(pv.row_ptr[p + 1] - r0);
    if (nrows == 0) return; // a patch whose dofs all belong to earlier patches writes nothing
    const int64_t n0 = pv.node_ptr[p];
    const int nnodes = (int)(pv.node_ptr[p + 1] - n0);
    double *accK = lds;
    double *accM = lds + (WK && WM ? pv.kcap : 0);
    RowDesc *desc = (RowDesc *)(lds + (WK && WM ? 2 : 1) * pv.kcap);
    int32_t *nid = (int32_t *)(desc + pv.max_rows);

    // the first instance's metadata travels while the LDS block is set up
    int64_t e = e0 + tid;
    uint4 lnv = make_uint4(0, 0, 0, 0);
    uint32_t sig = 0;
    if (e < e1) { lnv = ((const uint4 *)pv.elem_ln)[e]; sig = pv.elem_sig[e]; }
    const RowDesc last = pv.row_desc[r0 + nrows - 1];
    const int nacc2 = (int)((last.off + last.len + 1) >> 1);
    for (int k = tid; k < nnodes; k += T) nid[k] = pv.pnode[n0 + k];
    for (int s = tid; s < nrows; s += T) desc[s] = pv.row_desc[r0 + s];
    {
        double2 *zK = (double2 *)accK, *zM = (double2 *)accM;
        for (int k = tid; k < nacc2; k += T) {
            zK[k] = make_double2(0.0, 0.0);
            if (WK && WM) zM[k] = make_double2(0.0, 0.0);
        }
    }
    __syncthreads();

    while (e < e1) {
        const int64_t en = e + T;
        uint4 lnn = make_uint4(0, 0, 0, 0);
        uint32_t sign = 0;
        if (en < e1) { lnn = ((const uint4 *)pv.elem_ln)[en]; sign = pv.elem_sig[en]; }
        uint32_t ln[8] = {lnv.x & 0xffffu, lnv.x >> 16, lnv.y & 0xffffu, lnv.y >> 16, lnv.z & 0xffffu, lnv.z >> 16, lnv.w & 0xffffu, lnv.w >> 16};
        double x[8][3];
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            const double *px = m.xyz + 3 * (int64_t)nid[ln[a]];
            x[a][0] = px[0]; x[a][1] = px[1]; x[a][2] = px[2];
        }
        uint4 cp[4];
        {
            const uint4 *cpp = (const uint4 *)(pv.sigtab + (size_t)sig * 64);
#pragma unroll
            for (int k = 0; k < 4; ++k) cp[k] = cpp[k];
        }
        uint32_t ro[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) ro[i] = ln[i] < (uint32_t)nrows ? desc[ln[i]].off : 0xFFFFFFFFu;
        int64_t cell = 0;
        if constexpr (FK || FM) cell = pv.elem_cell[e];
        hex8_instance<WK, WM, FK, FM>(x, [&](uint4(&c4)[4], uint32_t(&r8)[8]) { for (int k = 0; k < 4; ++k) c4[k] = cp[k]; for (int k = 0; k < 8; ++k) r8[k] = ro[k]; }, cell, faK, faM, accK, accM, st, pv.elem_cell, e);
        e = en; lnv = lnn; sig = sign;
    }
    __syncthreads();
#ifdef TB_ABLATION
    if (faK.debug & 2) { if (tid == 0 && accK[0] + accM[1] == 1.2345) nzK[p] = 0.0; return; }
#endif

    // write-out: one row per half-wave, each matrix's row as one contiguous run
    const int half = tid >> 5, hl = tid & 31, nhalves = T >> 5;
#pragma unroll 2
    for (int s = half; s < nrows; s += nhalves) {
        const RowDesc d = desc[s];
        for (uint32_t k = hl; k < d.len; k += 32) {
            if constexpr (WK) nzK[d.nz0 + k] = accK[d.off + k];
            if constexpr (WM) nzM[d.nz0 + k] = accM[d.off + k];
        }
    }
}

// Default form: one workgroup per patch.  The patch header is one 16-byte scalar load; instance metadata, row descriptors and the
// pre-gathered vertex coordinates of the patch's nodes (contiguous, coalesced) are all requested at once right behind it, so a patch costs two
// dependent trips to memory (header → inputs) instead of four (pointers → last row descriptor → node ids → coordinate gather).
template <bool WK, bool WM, bool FK, bool FM, bool DIAG = false>
__global__ void __launch_bounds__(256, 2)
k_patch_hex8_staged(FormArgs faK, FormArgs faM, FusedView pv, const uint4 *__restrict__ hdrs, const double *__restrict__ pcoord, double *__restrict__ nzK,
                    double *__restrict__ nzM, Status *st)
{
    extern __shared__ double lds[];
    constexpr int T = 256;
    const int tid = threadIdx.x;
#ifdef TB_ABLATION
#define TB_ST(k) do { if (pv.prof && tid == 0 && (blockIdx.x & 1023) == 7) pv.prof[(blockIdx.x >> 10) * 8 + (k)] = wall_clock64(); } while (0)
#else
#define TB_ST(k) do { } while (0)
#endif
    TB_ST(0);
    const uint4 h = hdrs[blockIdx.x];
    const int64_t e0 = h.x, r0 = h.y, n0 = h.z;
    const int nrows = (int)(h.w & 0x3ff), nnodes = (int)((h.w >> 10) & 0x7ff), ne = (int)(h.w >> 21);
    if (nrows == 0) return;
    double *accK = lds;
    double *accM = lds + (WK && WM ? pv.kcap : 0);
    RowDesc *desc = (RowDesc *)(lds + (WK && WM ? 2 : 1) * pv.kcap);
    double *xs = (double *)(desc + pv.max_rows);
    uint4 lnv = make_uint4(0, 0, 0, 0);
    uint32_t sig = 0;
    if (tid < ne) { lnv = ((const uint4 *)pv.elem_ln)[e0 + tid]; sig = pv.elem_sig[e0 + tid]; }
    // every input of the patch is requested before anything waits (a plain copy loop would take one trip to memory per iteration)
    constexpr int NX = 5;
    uint4 rd = make_uint4(0, 0, 0, 0); // a RowDesc as four dwords (a struct temporary would live in scratch memory, and its store would wait for the load)
    double xc[NX];
    if (tid < nrows) rd = ((const uint4 *)pv.row_desc)[r0 + tid];
    {
        const double *pc = pcoord + 3 * n0;
#pragma unroll
        for (int j = 0; j < NX; ++j) xc[j] = tid + j * T < 3 * nnodes ? pc[tid + j * T] : 0.0;
    }
    {   // zero both accumulator blocks while the loads travel
        double2 *z = (double2 *)lds;
        const int n2 = ((WK && WM ? 2 : 1) * pv.kcap) >> 1;
        for (int k = tid; k < n2; k += T) z[k] = make_double2(0.0, 0.0);
    }
    if (tid < nrows) ((uint4 *)desc)[tid] = rd;
    for (int s = tid + T; s < nrows; s += T) desc[s] = pv.row_desc[r0 + s];                   // oversize patches only
#pragma unroll
    for (int j = 0; j < NX; ++j) if (tid + j * T < 3 * nnodes) xs[tid + j * T] = xc[j];
    for (int k = tid + NX * T; k < 3 * nnodes; k += T) xs[k] = pcoord[3 * n0 + k];            // oversize patches only
    uint4 cp[4];
    if (tid < ne) {
        const uint4 *cpp = (const uint4 *)(pv.sigtab + (size_t)sig * 64);
#pragma unroll
        for (int k = 0; k < 4; ++k) cp[k] = cpp[k];
    }
    TB_ST(1);
    __syncthreads();
    TB_ST(2);
    for (int ei = tid; ei < ne; ei += T) {
        if (ei >= T) {
            lnv = ((const uint4 *)pv.elem_ln)[e0 + ei];
            const uint4 *cpp = (const uint4 *)(pv.sigtab + (size_t)pv.elem_sig[e0 + ei] * 64);
#pragma unroll
            for (int k = 0; k < 4; ++k) cp[k] = cpp[k];
        }
        const uint32_t ln[8] = {lnv.x & 0xffffu, lnv.x >> 16, lnv.y & 0xffffu, lnv.y >> 16, lnv.z & 0xffffu, lnv.z >> 16, lnv.w & 0xffffu, lnv.w >> 16};
        double x[8][3];
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            const double *px = xs + 3 * ln[a];
            x[a][0] = px[0]; x[a][1] = px[1]; x[a][2] = px[2];
        }
        uint32_t ro[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) ro[i] = ln[i] < (uint32_t)nrows ? desc[ln[i]].off : 0xFFFFFFFFu;
        int64_t cell = 0;
        if constexpr (FK || FM) cell = pv.elem_cell[e0 + ei];
        hex8_instance<WK, WM, FK, FM, DIAG>(x, [&](uint4(&c4)[4], uint32_t(&r8)[8]) { for (int k = 0; k < 4; ++k) c4[k] = cp[k]; for (int k = 0; k < 8; ++k) r8[k] = ro[k]; }, cell, faK, faM, accK, accM, st, pv.elem_cell, e0 + ei);
    }
    TB_ST(3);
    __syncthreads();
    TB_ST(4);
#ifdef TB_ABLATION
    if (faK.debug & 2) { if (tid == 0 && accK[0] + accM[1] == 1.2345) nzK[blockIdx.x] = 0.0; return; }
#endif
    // write-out: one row per half-wave, four rows of a half-wave in flight (descriptor reads, accumulator reads and stores grouped)
    const int half = tid >> 5, hl = tid & 31;
    for (int s0 = half; s0 < nrows; s0 += 4 * (T / 32)) {
        uint4 d[4]; // {nz0 lo, nz0 hi, off, len}
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int s = s0 + u * (T / 32); d[u] = ((const uint4 *)desc)[s < nrows ? s : s0]; if (s >= nrows) d[u].w = 0; }
        double vK[4], vM[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t a_ = d[u].z + (hl < d[u].w ? hl : 0);
            if constexpr (WK) vK[u] = accK[a_];
            if constexpr (WM) vM[u] = accM[a_];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (hl < d[u].w) {
                const int64_t g0 = (int64_t)(((uint64_t)d[u].y << 32) | d[u].x) + hl;
                if constexpr (WK) nzK[g0] = vK[u];
                if constexpr (WM) nzM[g0] = vM[u];
            }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            for (uint32_t k = hl + 32; k < d[u].w; k += 32) { // rows longer than 32 entries (not on hexahedral meshes)
                const int64_t g0 = (int64_t)(((uint64_t)d[u].y << 32) | d[u].x) + k;
                if constexpr (WK) nzK[g0] = accK[d[u].z + k];
                if constexpr (WM) nzM[g0] = accM[d[u].z + k];
            }
    }
    TB_ST(5);
#ifdef TB_ABLATION
    if (pv.prof) { __builtin_amdgcn_s_waitcnt(0); TB_ST(6); }
#endif
#undef TB_ST
}

// One-trip form (default for constant coefficients): every input of the patch sits in one fixed-stride record (PatchFusedPlan::d_rec), so all
// of its loads — counts word, node indices, signature ids, row descriptors, coordinates — are addressed from blockIdx alone and issued at once:
// ONE trip to memory in front of the integration instead of header → inputs (phase stamps of the staged kernel at 216³: 3.4 of a workgroup's
// 14.2 µs before its first barrier).  A look-ahead (one dword load per lane into the record `pf` patches ahead, so that the next workgroup's trip would end
// in the cache hierarchy) was built and measured slower at every distance tried (1.94 → 2.02–2.05 ms at 512 … 2 048 patches ahead); removed — its keep-alive held two registers through the integration, one of them spilled.
// Write-out: a half-wave reads ALL its row descriptors, then ALL its accumulators, then stores (one dependent LDS round instead of five).
// KOFF > 0: the mass block sits KOFF entries behind the stiffness block (compile-time: the second add of a pair reuses the first one's address register with
// an immediate offset); KOFF = 0: `kcap` entries behind it (patches with more than KOFF accumulator entries)
template <bool WK, bool WM, bool DIAG, bool ISO, int RPH, int KOFF = 0, bool WQ = true>
__global__ void __launch_bounds__(256, 2)
k_patch_hex8_record(FormArgs faK, FormArgs faM, const uint8_t *__restrict__ rec, int stride, int nem, int rm, int nm, int kcap, const uint8_t *__restrict__ sigtab, const int32_t *__restrict__ elem_cell, int pf, int prio, int stagger,
                    double *__restrict__ nzK, double *__restrict__ nzM, Status *st
#ifdef TB_ABLATION
                    , long long *prof
#endif
)
{
    extern __shared__ double lds[];
    constexpr int T = 256;
    const int tid = threadIdx.x;
    const int wave_s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef TB_ABLATION
#define TB_ST(k) do { if (prof && tid == 0 && (blockIdx.x & 1023) == 7) prof[(blockIdx.x >> 10) * 8 + (k)] = wall_clock64(); } while (0)
#else
#define TB_ST(k) do { } while (0)
#endif
    TB_ST(0);
    // staging and write-out are short, latency-bound instruction streams; the integration is ≈ 2 400 instructions per lane.  With two workgroups per
    // CU a wave in its staging or write-out phase shares its SIMD with a wave of the other workgroup that may be integrating: raised priority lets
    // the short phase issue its loads / stores at once instead of waiting behind the other wave's arithmetic (TB_PATCH_PRIO=0 switches it off)
    // prio ≥ 2 (profiling build): asymmetric residents — the workgroup in the CU's first LDS slot integrates at raised priority, its co-resident
    // fills the gaps (two waves integrating at half rate each finish together and then wait together; one after the other, the second's staging /
    // write-out hides behind the first's arithmetic)
    bool slot0 = false;
    if (prio >= 2) slot0 = (__builtin_amdgcn_s_getreg((31 << 11) | 6) & 0xfffu) == 0u; // HW_REG_LDS_ALLOC: base of this workgroup's LDS
    if (prio == 1 || prio == 2) __builtin_amdgcn_s_setprio(3);
    // first generation only: the second workgroup of every CU (blocks 256…511 under round-robin dispatch) starts `stagger` × 64 cycles late, so that the
    // two residents of a CU run their integration phases in turn instead of together (uniform patches keep the offset once it is there)
    if (stagger > 0 && blockIdx.x >= 256 && blockIdx.x < 512)
        for (int k = 0; k < stagger; k += 100) __builtin_amdgcn_s_sleep(100);
    const uint8_t *r = rec + (size_t)blockIdx.x * (size_t)stride;
    const uint4 h4 = *(const uint4 *)r; // wave-uniform address: a scalar load, consumed only behind the vector loads below
    const uint32_t hw = h4.x;
    (void)pf; // (look-ahead touch of the record pf patches ahead: measured slower, its keep-alive cost the kernel two registers — removed)
    uint4 lnv = ((const uint4 *)(r + 16))[tid];
    const uint32_t sig = ((const uint32_t *)(r + 16 + (size_t)nem * 16))[tid];
    const uint8_t *rdp = r + 16 + (size_t)nem * 20;
    const double *pc = (const double *)(rdp + (size_t)rm * 16);
    constexpr int NX = ISO ? 6 : 5;
    uint4 rd = make_uint4(0, 0, 0, 0);
    double xc[NX];
    if (tid < rm) rd = ((const uint4 *)rdp)[tid];
    if constexpr (ISO) { // node-wise: lane t takes nodes t and t + 256 whole, so that it can map them (x′ = L⁻¹x) on their way to LDS
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int d = 0; d < 3; ++d) xc[3 * j + d] = tid + j * T < nm ? pc[3 * (tid + j * T) + d] : 0.0;
    } else {
#pragma unroll
        for (int j = 0; j < NX; ++j) xc[j] = tid + j * T < 3 * nm ? pc[tid + j * T] : 0.0;
    }
    const int kblk = KOFF > 0 ? KOFF : kcap;
    double *accK = lds;
    double *accM = lds + (WK && WM ? kblk : 0);
    RowDesc *desc = (RowDesc *)(lds + (WK && WM ? 2 : 1) * kblk);
    double *xs = (double *)(desc + rm);
    {   // zero both accumulator blocks while the loads travel
        double2 *z = (double2 *)lds;
        const int n2 = ((WK && WM ? 2 : 1) * kblk) >> 1;
        for (int k = tid; k < n2; k += T) z[k] = make_double2(0.0, 0.0);
    }
    if (tid < rm) ((uint4 *)desc)[tid] = rd;
    if constexpr (ISO) {
        const double *Li = faK.Linv;
        auto put = [&](int node, double x0, double x1, double x2) {
            xs[3 * node] = Li[0] * x0 + Li[1] * x1 + Li[2] * x2; xs[3 * node + 1] = Li[3] * x0 + Li[4] * x1 + Li[5] * x2; xs[3 * node + 2] = Li[6] * x0 + Li[7] * x1 + Li[8] * x2;
        };
#pragma unroll
        for (int j = 0; j < 2; ++j) if (tid + j * T < nm) put(tid + j * T, xc[3 * j], xc[3 * j + 1], xc[3 * j + 2]);
        for (int k = tid + 2 * T; k < nm; k += T) put(k, pc[3 * k], pc[3 * k + 1], pc[3 * k + 2]); // more than 512 nodes per patch
    } else {
#pragma unroll
        for (int j = 0; j < NX; ++j) if (tid + j * T < 3 * nm) xs[tid + j * T] = xc[j];
        for (int k = tid + NX * T; k < 3 * nm; k += T) xs[k] = pc[k]; // more than 426 nodes per patch
    }
    uint4 cp[4];
    {
        const uint4 *cpp = (const uint4 *)(sigtab + (size_t)sig * 64); // padding lanes carry signature 0: a valid read
#pragma unroll
        for (int k = 0; k < 4; ++k) cp[k] = cpp[k];
    }
    const int nrows = (int)(hw & 0x3ff), ne = (int)(hw >> 21);
    TB_ST(1);
    __syncthreads();
    TB_ST(2);
    if (prio == 1) __builtin_amdgcn_s_setprio(0);
    else if (prio >= 2) { if (!slot0) __builtin_amdgcn_s_setprio(0); else if (prio == 4) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(3); }
    for (int ei = tid; ei < ne; ei += T) { // one pass for all but the domain-boundary patches (they own the extra boundary layers: up to ≈ 280 instances)
        if (ei >= T) {
            lnv = ((const uint4 *)(r + 16))[ei];
            const uint4 *cpp = (const uint4 *)(sigtab + (size_t)((const uint32_t *)(r + 16 + (size_t)nem * 16))[ei] * 64);
#pragma unroll
            for (int k = 0; k < 4; ++k) cp[k] = cpp[k];
        }
        const uint32_t ln[8] = {lnv.x & 0xffffu, lnv.x >> 16, lnv.y & 0xffffu, lnv.y >> 16, lnv.z & 0xffffu, lnv.z >> 16, lnv.w & 0xffffu, lnv.w >> 16};
        double x[8][3];
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            const double *px = xs + 3 * ln[a];
            x[a][0] = px[0]; x[a][1] = px[1]; x[a][2] = px[2];
        }
        uint32_t ro[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) ro[i] = ln[i] < (uint32_t)nrows ? desc[ln[i]].off : 0xFFFFFFFFu;
        hex8_instance<WK, WM, false, false, DIAG, ISO>(x, [&](uint4(&c4)[4], uint32_t(&r8)[8]) { for (int k = 0; k < 4; ++k) c4[k] = cp[k]; for (int k = 0; k < 8; ++k) r8[k] = ro[k]; }, 0, faK, faM, accK, accM, st,
                                                  elem_cell, (int64_t)h4.y + ei);
    }
    TB_ST(3);
    __syncthreads();
    TB_ST(4);
    if (prio == 1 || prio == 2) __builtin_amdgcn_s_setprio(3);
    // write-out: one row per half-wave; all descriptors, then all accumulators, then all stores of the half-wave's rows.  (The thread index is
    // formed again from the lane count and the wave's number — a scalar — instead of being kept through the integration, where all 256 registers
    // are taken: kept, it was the kernel's one spilled value.)
    const int tidw = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) + 64 * wave_s;
    if constexpr (WQ) {
        // round 6: one row per QUARTER-wave, two entries per lane (rows of ≤ 32 entries; 14 lanes carry a hexahedral row's 27) — half the LDS reads and
        // global stores of the one-entry-per-lane form below (kept for rows longer than 32 entries).  All descriptors, then all accumulators, then all
        // stores, in two rounds of RPH / 4 rows per quarter-wave.
        const int quarter = tidw >> 4, e2 = (tidw & 15) << 1;
        constexpr int RQ = RPH / 2, RH = RQ / 2;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint4 d[RH]; // {nz0 lo, nz0 hi, off, len}
#pragma unroll
            for (int u = 0; u < RH; ++u) { const int s = quarter + (h * RH + u) * (T / 16); d[u] = ((const uint4 *)desc)[s < nrows ? s : 0]; }
            asm volatile("" ::: "memory"); // all descriptor reads in flight before the first is looked at
#pragma unroll
            for (int u = 0; u < RH; ++u) if (quarter + (h * RH + u) * (T / 16) >= nrows) d[u].w = 0;
            double vK[RH][2], vM[RH][2];
#pragma unroll
            for (int u = 0; u < RH; ++u) {
                const uint32_t a_ = d[u].z + ((uint32_t)e2 < d[u].w ? e2 : 0);
                if constexpr (WK) { vK[u][0] = accK[a_]; vK[u][1] = accK[a_ + 1]; } // (an odd row's last lane reads one slot past the row: inside the LDS allocation, unused)
                if constexpr (WM) { vM[u][0] = accM[a_]; vM[u][1] = accM[a_ + 1]; }
            }
#pragma unroll
            for (int u = 0; u < RH; ++u) {
                const int64_t g0 = (int64_t)(((uint64_t)d[u].y << 32) | d[u].x) + e2;
                if ((uint32_t)e2 + 1 < d[u].w) { // 16-byte stores at 8-byte aligned addresses (global memory takes them)
                    if constexpr (WK) *(double2 *)(nzK + g0) = make_double2(vK[u][0], vK[u][1]);
                    if constexpr (WM) *(double2 *)(nzM + g0) = make_double2(vM[u][0], vM[u][1]);
                } else if ((uint32_t)e2 < d[u].w) {
                    if constexpr (WK) nzK[g0] = vK[u][0];
                    if constexpr (WM) nzM[g0] = vM[u][0];
                }
            }
#pragma unroll
            for (int u = 0; u < RH; ++u)
                for (uint32_t k2 = 32 + ((uint32_t)e2 >> 1); k2 < d[u].w; k2 += 16) { // rows longer than 32 entries (nodes of an unstructured mesh with more than 8 cells)
                    const int64_t g0 = (int64_t)(((uint64_t)d[u].y << 32) | d[u].x) + k2;
                    if constexpr (WK) nzK[g0] = accK[d[u].z + k2];
                    if constexpr (WM) nzM[g0] = accM[d[u].z + k2];
                }
        }
        for (int s = quarter + RQ * (T / 16); s < nrows; s += T / 16) { // patches with more than 8·RPH rows
            const uint4 dd = ((const uint4 *)desc)[s];
            const int64_t g0 = (int64_t)(((uint64_t)dd.y << 32) | dd.x);
            for (uint32_t k = (uint32_t)e2 >> 1; k < dd.w; k += 16) {
                if constexpr (WK) nzK[g0 + k] = accK[dd.z + k];
                if constexpr (WM) nzM[g0 + k] = accM[dd.z + k];
            }
        }
    } else {
    const int half = tidw >> 5, hl = tidw & 31;
    uint4 d[RPH]; // {nz0 lo, nz0 hi, off, len}
#pragma unroll
    for (int u = 0; u < RPH; ++u) { const int s = half + u * (T / 32); d[u] = ((const uint4 *)desc)[s < nrows ? s : 0]; if (s >= nrows) d[u].w = 0; }
    double vK[RPH], vM[RPH];
#pragma unroll
    for (int u = 0; u < RPH; ++u) {
        const uint32_t a_ = d[u].z + (hl < d[u].w ? hl : 0);
        if constexpr (WK) vK[u] = accK[a_];
        if constexpr (WM) vM[u] = accM[a_];
    }
#pragma unroll
    for (int u = 0; u < RPH; ++u)
        if (hl < d[u].w) {
            const int64_t g0 = (int64_t)(((uint64_t)d[u].y << 32) | d[u].x) + hl;
            if constexpr (WK) nzK[g0] = vK[u];
            if constexpr (WM) nzM[g0] = vM[u];
        }
    for (int s = half + RPH * (T / 32); s < nrows; s += T / 32) { // patches with more rows than RPH per half-wave
        const uint4 dd = ((const uint4 *)desc)[s];
        const int64_t g0 = (int64_t)(((uint64_t)dd.y << 32) | dd.x);
        for (uint32_t k = hl; k < dd.w; k += 32) {
            if constexpr (WK) nzK[g0 + k] = accK[dd.z + k];
            if constexpr (WM) nzM[g0 + k] = accM[dd.z + k];
        }
    }
#pragma unroll
    for (int u = 0; u < RPH; ++u)
        for (uint32_t k = hl + 32; k < d[u].w; k += 32) { // rows longer than 32 entries (not on hexahedral meshes)
            const int64_t g0 = (int64_t)(((uint64_t)d[u].y << 32) | d[u].x) + k;
            if constexpr (WK) nzK[g0] = accK[d[u].z + k];
            if constexpr (WM) nzM[g0] = accM[d[u].z + k];
        }
    }
    TB_ST(5);
#ifdef TB_ABLATION
    if (prof) { __builtin_amdgcn_s_waitcnt(0); TB_ST(6); }
#endif
#undef TB_ST
}

// Round 6 built a PERSISTENT form of this kernel (k_patch_hex8_stream: two workgroups per CU looping over ticket-dealt patches, the record of patch k + 1
// requested at the start of patch k's write-out and landed in LDS behind it, accumulators zeroed by the lane that reads them out, quarter-wave
// write-out) and removed it again: parity-green, and 2–4 % SLOWER at 216³ (1.66–1.67 against 1.60–1.62 ms on the same boxes) and 14 % slower on the
// 27-layer slab.  Its phase stamps showed why the staging trip is not what this kernel waits for: with the trip hidden the integration phase grew from
// 6.6 to 7.3 µs (the co-resident workgroup is then integrating more of the time) and the write-out, now always beside an integrating partner whose
// ds_add_f64 stream keeps the LDS pipe busy, from 2.5 to 4.6 µs.  The work-skipping builds put the floor where it is: 1.22 ms with neither arithmetic
// nor LDS adds (5.7 GB at 4.7 TB/s — the rate of a copy on this part), +0.21 ms for the arithmetic alone, +0.25 ms for the adds alone, 1.65 ms with
// both.  Logs: profiles/r06_v1/ab_stream_*.log, ab_work_skipping_flags.log; what was kept from it is the quarter-wave write-out (WQ above).

bool hex8_patch_applicable(const tb_form *f, const tb_pattern *p)
{
    static const bool legacy = getenv("TB_PATCH_KERNEL") && !strcmp(getenv("TB_PATCH_KERNEL"), "legacy");
    if (legacy) return false;
    const tb_mesh *m = f->mesh;
    if (m->field_kind != TB_HEX8 || m->geom_kind != TB_HEX8 || m->ncomp != 1 || f->qorder != 2) return false;
    if (f->kind == TB_FORM_DIFFUSION && !f->symmetric) return false;
    if (f->kind != TB_FORM_DIFFUSION && f->kind != TB_FORM_MASS) return false;
    if (f->has_cellset) return false;
    (void)p;
    return true;
}

int tabulate_diffusion_field(tb_form *f); // tb_assembly.hip

// G_q = −¼/detJ·A·D_q·Aᵀ and detJ_q of every cell for a diffusion form with a field tensor, built once per form (mesh and field are fixed)
int tabulate_g_hex8(tb_form *fK)
{
    tb_mesh *m = fK->mesh;
    tb_device *dev = m->dev;
    int rc;
    if (!fK->d_dtab) { rc = tabulate_diffusion_field(fK); if (rc) return rc; }
    if (fK->d_gtab) return TB_OK;
    const size_t bytes = sizeof(double) * 56 * (size_t)m->n_cells;
    hipError_t e = hipMalloc((void **)&fK->d_gtab, bytes);
    if (e != hipSuccess) { set_error("G table of the hexahedron patch kernel (%zu B): %s", bytes, hipGetErrorString(e)); return TB_ERR_NOMEM; }
    hipLaunchKernelGGL(k_tabulate_g_hex8, dim3((unsigned)((m->n_cells + 255) / 256)), dim3(256), 0, dev->stream, make_view(m), fK->d_dtab, fK->d_gtab, dev->d_status);
    TB_HIP(hipGetLastError());
    rc = check_status(dev);
    if (rc) { (void)hipFree(fK->d_gtab); fK->d_gtab = nullptr; return rc; }
    return TB_OK;
}

int launch_assemble_hex8_patch(tb_form *fK, tb_form *fM, tb_pattern *p, double t, double *d_nzK, double *d_nzM)
{
    tb_form *any = fK ? fK : fM;
    tb_mesh *m = any->mesh;
    tb_device *dev = m->dev;
    int rc = reset_status(dev);
    if (rc) return rc;
    if (fK && fK->field) { rc = tabulate_g_hex8(fK); if (rc) return rc; }
    const int nreg = fK && fM ? 2 : 1;
    rc = ensure_patch_fused(m, p, nreg);
    if (rc) return rc;
    const PatchPlan *pp = m->patches.get();
    const PatchFusedPlan *pf = p->patch_fused.get();
    FusedView pv{};
    pv.elem_ptr = pp->d_elem_ptr; pv.row_ptr = pp->d_row_ptr; pv.node_ptr = pf->d_node_ptr; pv.elem_cell = pp->d_elem_cell;
    pv.elem_ln = pf->d_elem_ln; pv.elem_sig = pf->d_elem_sig; pv.sigtab = pf->d_sigtab; pv.pnode = pf->d_pnode; pv.row_desc = pf->d_row_desc;
    pv.kcap = pf->max_lds_entries; pv.max_rows = pp->max_rows; pv.max_nodes = pf->max_nodes;
    // one LDS size for both kernel forms: accumulator block(s) + row descriptors + 3 doubles per patch node (the general form keeps 4-byte node ids there)
    const size_t lds = (size_t)nreg * pv.kcap * sizeof(double) + (size_t)pv.max_rows * sizeof(RowDesc) + (size_t)pv.max_nodes * 3 * sizeof(double);
    FormArgs aK = fK ? make_args(fK, t) : FormArgs{};
    const FormArgs aM = fM ? make_args(fM, t) : FormArgs{};
#ifdef TB_ABLATION
    aK.debug = aM.debug | aK.debug;
    static long long *d_prof = nullptr; // TB_PROF_STAMPS: phase time stamps of every 1024th workgroup
    const int nprof = (int)(pp->n_patches >> 10) + 1;
    if (!d_prof && getenv("TB_PROF_STAMPS")) TB_HIP(hipMalloc((void **)&d_prof, (size_t)nprof * 8 * sizeof(long long)));
    if (d_prof) TB_HIP(hipMemsetAsync(d_prof, 0, (size_t)nprof * 8 * sizeof(long long), dev->stream));
    pv.prof = d_prof;
#endif
    const bool fk = fK && fK->field, fm = fM && fM->field;
    static const bool general = getenv("TB_PATCH_KERNEL") && !strcmp(getenv("TB_PATCH_KERNEL"), "general"); // measured alternative
    const bool staged = pf->d_hdr && !general;
    const bool diag = fK && !fk && aK.D[1] == 0.0 && aK.D[2] == 0.0 && aK.D[5] == 0.0 && aK.D[3] == 0.0 && aK.D[6] == 0.0 && aK.D[7] == 0.0;
    // one-trip record kernel: constant coefficients, patches of ≤ 256 instances (TB_PATCH_KERNEL=staged keeps the two-trip kernel for A/B runs)
    static const bool no_record = getenv("TB_PATCH_KERNEL") && strcmp(getenv("TB_PATCH_KERNEL"), "record") != 0;
    static const int wave_prio = tune_env("TB_PATCH_PRIO") ? atoi(tune_env("TB_PATCH_PRIO")) : 0; // measured: no effect (1.7007 vs 1.6997 ms)
    static const int stagger = tune_env("TB_PATCH_STAGGER") ? atoi(tune_env("TB_PATCH_STAGGER")) : 0;
    const int pf_ahead = 0; // (look-ahead touch of later records: removed from the record kernel, see there)
    if (staged && !no_record && !fk && !fm && ensure_patch_records(p) == TB_OK) {
        PatchFusedPlan *pr = p->patch_fused.get();
        constexpr int KOFF = 4096; // fused pair: mass block at a fixed distance when it fits (150 rows of 27 entries = 4 050)
        const bool fixm = fK && fM && pv.kcap <= KOFF &&
                          (size_t)2 * KOFF * sizeof(double) + (size_t)pr->rec_rm * sizeof(RowDesc) + (size_t)pr->rec_nm * 3 * sizeof(double) <= 80 * 1024;
        const size_t ldsr = (size_t)nreg * (fixm ? KOFF : pv.kcap) * sizeof(double) + (size_t)pr->rec_rm * sizeof(RowDesc) + (size_t)pr->rec_nm * 3 * sizeof(double);
        if (ldsr <= 80 * 1024) {
            constexpr int RPH = 20; // rows per half-wave kept in registers at write-out (160 rows per patch; more take the loop)
            // constant positive definite tensor: D = L·Lᵀ, the kernel integrates in the coordinates x′ = L⁻¹x where the tensor is the identity
            // (TB_PATCH_ISO=0 keeps the tensor products for A/B runs); anything else — indefinite, or the switch — takes the DIAG / general instances
            static const bool iso_on = !(getenv("TB_PATCH_ISO") && atoi(getenv("TB_PATCH_ISO")) == 0);
            bool iso = false;
            FormArgs aMi = aM;
            if (fK && iso_on) {
                const double *D = aK.D;
                double L[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
                bool pd = D[1] == D[3] && D[2] == D[6] && D[5] == D[7];
                for (int i = 0; i < 3 && pd; ++i)
                    for (int j = 0; j <= i; ++j) {
                        double sum = D[3 * i + j];
                        for (int k2 = 0; k2 < j; ++k2) sum -= L[i][k2] * L[j][k2];
                        if (i == j) { if (!(sum > 0.0)) { pd = false; break; } L[i][i] = std::sqrt(sum); }
                        else L[i][j] = sum / L[j][j];
                    }
                if (pd) {
                    double Li[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}; // inverse of the lower triangle
                    for (int i = 0; i < 3; ++i) {
                        Li[i][i] = 1.0 / L[i][i];
                        for (int j = 0; j < i; ++j) { double sum = 0.0; for (int k2 = j; k2 < i; ++k2) sum -= L[i][k2] * Li[k2][j]; Li[i][j] = sum / L[i][i]; }
                    }
                    const double detL = L[0][0] * L[1][1] * L[2][2];
                    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) aK.Linv[3 * i + j] = Li[i][j];
                    aK.iso_scale = -0.25 * detL;
                    aMi.rho = aM.rho * detL; // the kernel's determinants are those of the mapped coordinates: det J = det L · det J′
                    iso = true;
                }
            }
            static const size_t lds_pad = tune_env("TB_PATCH_LDS_PAD") ? (size_t)atol(tune_env("TB_PATCH_LDS_PAD")) : 0; // profiling build: > 80 KB leaves one workgroup per CU
            const size_t ldsr_ = ldsr;
            auto launch_rec = [&](auto k) -> int {
                const size_t ldsr = std::max(ldsr_, lds_pad);
                TB_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsr));
#ifdef TB_ABLATION
                hipLaunchKernelGGL(k, dim3((unsigned)pp->n_patches), dim3(256), ldsr, dev->stream, aK, aMi, (const uint8_t *)pr->d_rec, pr->rec_stride, pr->rec_ne, pr->rec_rm, pr->rec_nm, pv.kcap,
                                   (const uint8_t *)pf->d_sigtab, pp->d_elem_cell, pf_ahead, wave_prio, stagger, d_nzK, d_nzM, dev->d_status, d_prof);
#else
                hipLaunchKernelGGL(k, dim3((unsigned)pp->n_patches), dim3(256), ldsr, dev->stream, aK, aMi, (const uint8_t *)pr->d_rec, pr->rec_stride, pr->rec_ne, pr->rec_rm, pr->rec_nm, pv.kcap,
                                   (const uint8_t *)pf->d_sigtab, pp->d_elem_cell, pf_ahead, wave_prio, stagger, d_nzK, d_nzM, dev->d_status);
#endif
                return TB_OK;
            };
            set_last_kernel("k_patch_hex8_record<%s,%s,RPH%d%s>", fK && fM ? "K+M" : fK ? "K" : "M", !fK ? "-" : iso ? "ISO" : diag ? "DIAG" : "GEN", RPH, fK && fM && fixm ? ",KOFF4096" : "");
            static const bool wq_off = tune_env("TB_PATCH_WQ") && atoi(tune_env("TB_PATCH_WQ")) == 0; // profiling build: the half-wave write-out of rounds 2-5 (bench instance only)
            if (fK && fM && fixm && iso && wq_off) rc = launch_rec(k_patch_hex8_record<true, true, false, true, RPH, KOFF, false>);
            else if (fK && fM && fixm) rc = iso ? launch_rec(k_patch_hex8_record<true, true, false, true, RPH, KOFF>) : diag ? launch_rec(k_patch_hex8_record<true, true, true, false, RPH, KOFF>) : launch_rec(k_patch_hex8_record<true, true, false, false, RPH, KOFF>);
            else if (fK && fM) rc = iso ? launch_rec(k_patch_hex8_record<true, true, false, true, RPH>) : diag ? launch_rec(k_patch_hex8_record<true, true, true, false, RPH>) : launch_rec(k_patch_hex8_record<true, true, false, false, RPH>);
            else if (fK) rc = iso ? launch_rec(k_patch_hex8_record<true, false, false, true, RPH>) : diag ? launch_rec(k_patch_hex8_record<true, false, true, false, RPH>) : launch_rec(k_patch_hex8_record<true, false, false, false, RPH>);
            else rc = launch_rec(k_patch_hex8_record<false, true, false, false, RPH>);
            if (rc) return rc;
            TB_HIP(hipGetLastError());
            goto done;
        }
    }
    {
    const MeshView mv = make_view(m);
    auto launch = [&](auto ks, auto kg) -> int {
        if (staged) {
            TB_HIP(hipFuncSetAttribute((const void *)ks, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(ks, dim3((unsigned)pp->n_patches), dim3(256), lds, dev->stream, aK, aM, pv, (const uint4 *)pf->d_hdr, pf->d_pcoord, d_nzK, d_nzM, dev->d_status);
        } else {
            TB_HIP(hipFuncSetAttribute((const void *)kg, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kg, dim3((unsigned)pp->n_patches), dim3(pp->threads), lds, dev->stream, mv, aK, aM, pv, d_nzK, d_nzM, dev->d_status);
        }
        return TB_OK;
    };
    set_last_kernel("%s<%s,%s%s%s>", staged ? "k_patch_hex8_staged" : "k_patch_hex8", fK && fM ? "K+M" : fK ? "K" : "M", !fK ? "-" : fk ? "FIELD" : diag ? "DIAG" : "GEN", fm ? ",RHO_FIELD" : "", "");
#define TB_PL(a, b, c, d) rc = launch(k_patch_hex8_staged<a, b, c, d>, k_patch_hex8<a, b, c, d>)
#define TB_PLD(a, b, c, d) rc = launch(k_patch_hex8_staged<a, b, c, d, true>, k_patch_hex8<a, b, c, d>)
    // constant diagonal tensors (isotropic / axis-aligned conductivities) take the variant with the cheaper A·D product
    if (fK && fM) { if (fk && fm) TB_PL(true, true, true, true); else if (fk) TB_PL(true, true, true, false); else if (fm) { if (diag) TB_PLD(true, true, false, true); else TB_PL(true, true, false, true); } else { if (diag) TB_PLD(true, true, false, false); else TB_PL(true, true, false, false); } }
    else if (fK) { if (fk) TB_PL(true, false, true, false); else if (diag) TB_PLD(true, false, false, false); else TB_PL(true, false, false, false); }
    else { if (fm) TB_PL(false, true, false, true); else TB_PL(false, true, false, false); }
#undef TB_PL
#undef TB_PLD
    if (rc) return rc;
    TB_HIP(hipGetLastError());
    }
done:
#ifdef TB_ABLATION
    if (d_prof && staged) { // average phase durations (µs; wall clock 100 MHz)
        std::vector<long long> h((size_t)nprof * 8);
        TB_SYNC_STREAM(dev);
        TB_HIP(hipMemcpy(h.data(), d_prof, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
        double ph[6] = {0, 0, 0, 0, 0, 0};
        int n = 0;
        for (int w = 0; w < nprof; ++w) {
            const long long *a = &h[(size_t)w * 8];
            if (!a[0] || !a[6]) continue;
            for (int k = 0; k < 6; ++k) ph[k] += (double)(a[k + 1] - a[k]) * 0.01;
            ++n;
        }
        if (n) fprintf(stderr, "[tbhip] staged phases (us): header+issue+zero %.2f | barrier(loads land) %.2f | lds-read+compute+adds %.2f | barrier %.2f | write-out issue %.2f | store drain %.2f  (n=%d)\n",
                       ph[0] / n, ph[1] / n, ph[2] / n, ph[3] / n, ph[4] / n, ph[5] / n, n);
    }
#endif
    return check_status(dev);
}

} // namespace tb
