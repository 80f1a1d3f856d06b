// tb_patch_sym.hip — PATCH-strategy assembly of the mass and diffusion matrices on trilinear hexahedra (2×2×2 Gauss rule) with
// SYMMETRIC row accumulators.  A measured alternative (TB_PATCH_KERNEL=sym), parity-green, NOT the default: on MI355X it is slower than
// k_patch_hex8_staged (tb_patch_fused.hip) — 2.6–3.0 ms against 1.95–2.09 ms for M + K at 216³ on the same boxes (profiles/r03_v1/sym_kernel.txt).
//
// Reference call sites: the backward-Euler heat stage assembles the mass and the diffusion operator of one DofHandler back to
// back on the same sparsity pattern (src/solver/time/euler.jl:172-176, pattern sharing :110-116); element routines
// src/modeling/core/mass.jl:28-43, src/modeling/core/diffusion.jl:28-50, geometry src/ferrite-addons/PR883.jl:253-291,367-387.
//
// The idea (round-2 review): the staged kernel issues 64 + 64 ds_add_f64 per cell instance × 1.70 instances per cell.  Both matrices are symmetric, so
//   * a coupling {i, j} of a cell is added ONCE: to the row of smaller patch-local slot among the rows the patch owns (36 instead of 64 adds per
//     matrix and instance), and the other triangle is filled when the rows leave — every CSR entry of an owned row knows the LDS entry that holds
//     it (its own run, or the run of the owned column's row) through a de-duplicated mirror map;
//   * a row's run holds ≈ 14 instead of 27 entries, so the two matrices can take turns on ONE accumulator block (stiffness pass, write-out, mass
//     pass, write-out; the mass pass repeats only the determinants of the geometry stage) with 7×7×7 tiles at two workgroups per CU — 1.51
//     instances per cell instead of 1.70 and 2.2× fewer patches to stage (SEQ); or keep one pass over two half-size blocks (!SEQ).
// What the phase stamps of the profiling build say (µs per 284-cell patch, SEQ): staging 4.0 | stiffness pass 12.5 (two sweeps of 256 instances:
// 6.2 each — the vector ALU's own time for ≈1 700 instructions per instance on two waves per SIMD, so the LDS adds were NOT what bounded the
// integration) | write-out 6.1 | mass pass 6.2 (≈700 instructions per instance, but a pass of its own: metadata, coordinates, determinants again) |
// write-out 5.6 = 36 µs, i.e. 127 ns per cell and workgroup slot against 93 for the staged kernel.  The mirrored write-out needs its map entries
// from memory: requested between the stores they wait for every earlier store (vmcnt retires in order and the compiler cannot count stores in
// divergent branches → vmcnt(0)), requested all at once before the stores they cost 48 LDS-descriptor reads + 48 loads per lane (4 µs);
// held in LDS (18 KB for a 7³ tile) they do not fit beside the accumulators, descriptors and coordinates.  Even with a free write-out the separate
// mass pass eats what the larger tile saves.  Kept for the record and for meshes / devices where the balance differs.
// Element arithmetic: tb_hex8_sumfac.hpp, unchanged.  Results differ from the staged kernel by summation order only.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

#include "tb_elem.hpp"
#include "tb_forms.hpp"
#include "tb_hex8_sumfac.hpp"
#include "tb_internal.h"

namespace tb {
using namespace tbk;

struct SymView {
    const int32_t *elem_cell;
    const uint4 *elem_ln;     // 8 patch-local node indices (16 bit) per instance
    const uint32_t *elem_ssig;
    const uint4 *ssigtab;     // 3 per signature: 36 rank bytes (pair (i ≤ j) in row-major order of the upper triangle) + padding
    const uint4 *row_sdesc;   // nz0 low | nz0 high 8 bits + length << 8 | run offset | mirror-map offset
    const int16_t *mirtab;
    int kcap, max_rows, max_nodes;
#ifdef TB_ABLATION
    long long *prof; // TB_PROF_STAMPS: 16 time stamps of every 256th workgroup
#endif
};

__device__ __forceinline__ constexpr int sym_pair_index(int i, int j) { return i * 8 - i * (i - 1) / 2 + (j - i); }

__device__ __forceinline__ uint32_t sym_word(const uint4 (&sg)[3], int k)
{
    const uint4 &q = sg[k >> 2];
    return (k & 3) == 0 ? q.x : (k & 3) == 1 ? q.y : (k & 3) == 2 ? q.z : q.w;
}

// One add per pair: target row = the one of smaller patch-local node index (owned nodes carry their row slot, the others larger numbers, so an
// owned row wins whenever there is one); ro = 0xFFFFFFFF marks "not owned".  val(i, j) is called with compile-time indices.
template <class ValFn>
__device__ __forceinline__ void sym_scatter(const uint32_t (&ln)[8], const uint32_t (&ro)[8], const uint4 (&sg)[3], double *acc, ValFn &&val)
{
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = i; j < 8; ++j) {
            const int pr = sym_pair_index(i, j);
            const uint32_t rk = (sym_word(sg, pr >> 2) >> (8 * (pr & 3))) & 0xffu;
            const uint32_t r = (i == j || ln[i] < ln[j]) ? ro[i] : ro[j];
            const double v = val(i, j);
            if (r != 0xFFFFFFFFu) unsafeAtomicAdd(acc + (r + rk), v);
        }
}

// SEQ: the two matrices take turns on one accumulator block (default); !SEQ: two blocks, one pass (measured alternative, smaller tiles).
template <bool WK, bool WM, bool FK, bool FM, bool DIAG, bool SEQ>
__global__ void __launch_bounds__(256, 2)
k_patch_hex8_sym(FormArgs faK, FormArgs faM, SymView pv, const uint4 *__restrict__ hdrs, const double *__restrict__ pcoord, double *__restrict__ nzK,
                 double *__restrict__ nzM, Status *st)
{
    extern __shared__ double lds[];
    constexpr int T = 256;
    constexpr bool TWO = WK && WM;
    const int tid = threadIdx.x;
#ifdef TB_ABLATION
#define TB_ST(k) do { if (pv.prof && tid == 0 && (blockIdx.x & 255) == 7) pv.prof[(blockIdx.x >> 8) * 16 + (k)] = wall_clock64(); } while (0)
#else
#define TB_ST(k) do { } while (0)
#endif
    TB_ST(0);
    const uint4 h = hdrs[blockIdx.x];
    const int64_t e0 = h.x, r0 = h.y, n0 = h.z;
    const int nrows = (int)(h.w & 0x3ff), nnodes = (int)((h.w >> 10) & 0x7ff), ne = (int)(h.w >> 21);
    if (nrows == 0) return;
    double *acc = lds;
    double *acc2 = lds + (TWO && !SEQ ? pv.kcap : 0);
    uint4 *desc = (uint4 *)(lds + (TWO && !SEQ ? 2 : 1) * pv.kcap);
    double *xs = (double *)(desc + pv.max_rows);

    // ---- staging: every input of the patch is requested before anything waits
    // instance metadata of the thread's first two instances (tid, tid + 256: the two sweeps of a full tile) stay in registers for both passes
    uint4 lnvA = make_uint4(0, 0, 0, 0), lnvB = make_uint4(0, 0, 0, 0);
    uint32_t sigA = 0, sigB = 0;
    if (tid < ne) { lnvA = pv.elem_ln[e0 + tid]; sigA = pv.elem_ssig[e0 + tid]; }
    if (tid + T < ne) { lnvB = pv.elem_ln[e0 + tid + T]; sigB = pv.elem_ssig[e0 + tid + T]; }
    constexpr int NX = 9, NRD = 2;
    uint4 rd[NRD];
    double xc[NX];
#pragma unroll
    for (int u = 0; u < NRD; ++u) rd[u] = tid + u * T < nrows ? pv.row_sdesc[r0 + tid + u * T] : make_uint4(0, 0, 0, 0);
    {
        const double *pc = pcoord + 3 * n0;
#pragma unroll
        for (int j = 0; j < NX; ++j) xc[j] = tid + j * T < 3 * nnodes ? pc[tid + j * T] : 0.0;
    }
    auto zero_acc = [&]() {
        double2 *z = (double2 *)lds;
        const int n2 = ((TWO && !SEQ ? 2 : 1) * pv.kcap) >> 1;
        for (int k = tid; k < n2; k += T) z[k] = make_double2(0.0, 0.0);
    };
    zero_acc();
#pragma unroll
    for (int u = 0; u < NRD; ++u) if (tid + u * T < nrows) desc[tid + u * T] = rd[u];
    for (int s = tid + NRD * T; s < nrows; s += T) desc[s] = pv.row_sdesc[r0 + s];            // oversize patches only
#pragma unroll
    for (int j = 0; j < NX; ++j) if (tid + j * T < 3 * nnodes) xs[tid + j * T] = xc[j];
    for (int k = tid + NX * T; k < 3 * nnodes; k += T) xs[k] = pcoord[3 * n0 + k];            // oversize patches only
    uint4 sgA[3], sgB[3];
    auto load_sig = [&](uint32_t sig, uint4 (&sg)[3]) {
        const uint4 *sp = pv.ssigtab + (size_t)sig * 3;
#pragma unroll
        for (int k = 0; k < 3; ++k) sg[k] = sp[k];
    };
    load_sig(sigA, sgA); // signature 0 exists whenever there is an instance: the unconditional loads are in range
    load_sig(sigB, sgB);
    TB_ST(1);
    __syncthreads();
    TB_ST(2);

    // ---- one pass over the instances; PASS_K / PASS_M select what is integrated (both: the one-pass form)
    auto integrate = [&](auto pass_k, auto pass_m) {
        constexpr bool PK = decltype(pass_k)::value, PM = decltype(pass_m)::value;
        for (int sw = 0; sw * T + tid < ne; ++sw) {
            const int ei = sw * T + tid;
            // sweep 0 / 1: the metadata kept in registers (component-wise selects on the wave-uniform sweep index: no private-memory array)
            auto pick = [&](const uint4 &a, const uint4 &b) { return make_uint4(sw ? b.x : a.x, sw ? b.y : a.y, sw ? b.z : a.z, sw ? b.w : a.w); };
            uint4 lnv = pick(lnvA, lnvB), sg[3] = {pick(sgA[0], sgB[0]), pick(sgA[1], sgB[1]), pick(sgA[2], sgB[2])};
            if (sw >= 2) { lnv = pv.elem_ln[e0 + ei]; load_sig(pv.elem_ssig[e0 + ei], sg); }   // oversize patches only
            const uint32_t ln[8] = {lnv.x & 0xffffu, lnv.x >> 16, lnv.y & 0xffffu, lnv.y >> 16, lnv.z & 0xffffu, lnv.z >> 16, lnv.w & 0xffffu, lnv.w >> 16};
            int64_t cell = 0;
            if constexpr (FK || FM) cell = pv.elem_cell[e0 + ei];
            double G[8][6], dq[8];
            constexpr bool NEED_X = !FK;
            if constexpr (NEED_X) {
                double x[8][3];
#pragma unroll
                for (int a = 0; a < 8; ++a) {
                    const double *px = xs + 3 * ln[a];
                    x[a][0] = px[0]; x[a][1] = px[1]; x[a][2] = px[2];
                }
                auto Dq = [&](int, double(&D)[6]) { D[0] = faK.D[0]; D[1] = faK.D[1]; D[2] = faK.D[2]; D[3] = faK.D[4]; D[4] = faK.D[5]; D[5] = faK.D[8]; };
                if (!hex8_sf_geometry<PK, false, DIAG>(x, Dq, G, dq)) { st->neg_detj = 1; st->cell = pv.elem_cell[e0 + ei]; }
            } else {
                // field tensors: G_q and detJ_q were tabulated once (k_tabulate_g_hex8; detJ ≤ 0 reported there)
                const double *gp = faK.gtab + cell * 56;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if constexpr (PK) {
#pragma unroll
                        for (int k = 0; k < 6; ++k) G[q][k] = gp[7 * q + k];
                    }
                    dq[q] = gp[7 * q + 6];
                }
            }
            Hex8SFK c;
            if constexpr (PK) hex8_sf_contract(G, c);
            uint32_t ro[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) ro[i] = ln[i] < (uint32_t)nrows ? desc[ln[i]].z : 0xFFFFFFFFu;
            if constexpr (PK) sym_scatter(ln, ro, sg, acc, [&](int i, int j) { return hex8_sf_entry(c, i, j); });
            if constexpr (PM) {
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
                sym_scatter(ln, ro, sg, PK ? acc2 : acc, [&](int i, int j) { return Z[SF::ty(i, j, 0)][SF::ty(i, j, 1)][SF::ty(i, j, 2)]; });
            }
        }
    };

    // ---- write-out with the mirror map.  Half-wave `half` owns rows half, half + 8, …  ALL its mirror-map entries (NR rounds of RW rows, one
    // 16-bit entry per lane and row) are requested in one go before the barrier that ends the integration — unconditional loads from clamped
    // addresses, no branch between them — so their latency hides behind the slower waves and no load ever queues behind the stores of the
    // write-out (vmcnt retires in order on gfx9: a map request issued between stores waits for every earlier store to be acknowledged, which
    // cost one store latency per round — measured 6 µs per write-out instead of 2.5).  Then the rows leave in rounds of RW rows per half-wave.
    constexpr int RW = 8, NR = 6;
    const int half = tid >> 5, hl = tid & 31;
    int16_t mm[NR][RW];
    auto request_maps = [&]() {
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int u = 0; u < RW; ++u) {
                const int s = half + 8 * (RW * j + u);
                const uint32_t *dp = (const uint32_t *)&desc[s < nrows ? s : nrows - 1];
                const uint32_t len = s < nrows ? (dp[1] >> 8) & 0xffu : 0u;
                mm[j][u] = pv.mirtab[(size_t)dp[3] + ((uint32_t)hl < len ? hl : 0)];
            }
    };
    auto write_out = [&](auto two, double *__restrict__ nzA, double *__restrict__ nzB) {
        constexpr bool BOTH = decltype(two)::value;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            if (8 * RW * j >= nrows) break;
            uint4 d[RW];
            double vA[RW], vB[RW];
#pragma unroll
            for (int u = 0; u < RW; ++u) {
                const int s = half + 8 * (RW * j + u);
                d[u] = desc[s < nrows ? s : nrows - 1];
                if (s >= nrows) d[u].y = 0; // length 0: nothing stored
            }
#pragma unroll
            for (int u = 0; u < RW; ++u) {
                const int a_ = (int)d[u].z + (int)mm[j][u];
                vA[u] = acc[a_];
                if constexpr (BOTH) vB[u] = acc2[a_];
            }
#pragma unroll
            for (int u = 0; u < RW; ++u) {
                const uint32_t len = (d[u].y >> 8) & 0xffu;
                const int64_t g0 = (int64_t)(((uint64_t)(d[u].y & 0xffu) << 32) | d[u].x);
                if ((uint32_t)hl < len) {
                    nzA[g0 + hl] = vA[u];
                    if constexpr (BOTH) nzB[g0 + hl] = vB[u];
                }
                if (len > 32) // rows longer than 32 entries (irregular nodes of unstructured hexahedral meshes)
                    for (uint32_t k = hl + 32; k < len; k += 32) {
                        const int a_ = (int)d[u].z + (int)pv.mirtab[(size_t)d[u].w + k];
                        nzA[g0 + k] = acc[a_];
                        if constexpr (BOTH) nzB[g0 + k] = acc2[a_];
                    }
            }
        }
        for (int s = half + 8 * RW * NR; s < nrows; s += 8) { // oversize patches only
            const uint4 d = desc[s];
            const uint32_t len = (d.y >> 8) & 0xffu;
            const int64_t g0 = (int64_t)(((uint64_t)(d.y & 0xffu) << 32) | d.x);
            for (uint32_t k = hl; k < len; k += 32) {
                const int a_ = (int)d.z + (int)pv.mirtab[(size_t)d.w + k];
                nzA[g0 + k] = acc[a_];
                if constexpr (BOTH) nzB[g0 + k] = acc2[a_];
            }
        }
    };
    using Yes = std::integral_constant<bool, true>;
    using No = std::integral_constant<bool, false>;

    if constexpr (TWO && !SEQ) {
        integrate(Yes{}, Yes{});
        TB_ST(3);
        request_maps();
        __syncthreads();
        TB_ST(4);
        write_out(Yes{}, nzK, nzM);
        TB_ST(5);
    } else if constexpr (TWO) {
        integrate(Yes{}, No{});
        TB_ST(3);
        request_maps();
        __syncthreads();
        TB_ST(4);
        write_out(No{}, nzK, nzK);
        TB_ST(5);
        __syncthreads();
        zero_acc();
        __syncthreads();
        TB_ST(6);
        integrate(No{}, Yes{});
        TB_ST(7);
        request_maps();
        __syncthreads();
        TB_ST(8);
        write_out(No{}, nzM, nzM);
        TB_ST(9);
    } else {
        if constexpr (WK) integrate(Yes{}, No{}); else integrate(No{}, Yes{});
        request_maps();
        __syncthreads();
        write_out(No{}, WK ? nzK : nzM, WK ? nzK : nzM);
    }
}

int tabulate_diffusion_field(tb_form *f);                                     // tb_assembly.hip
int tabulate_g_hex8(tb_form *fK);                                             // tb_patch_fused.hip

int launch_assemble_hex8_sym(tb_form *fK, tb_form *fM, tb_pattern *p, double t, double *d_nzK, double *d_nzM)
{
    tb_form *any = fK ? fK : fM;
    tb_mesh *m = any->mesh;
    tb_device *dev = m->dev;
    int rc = reset_status(dev);
    if (rc) return rc;
    if (fK && fK->field) { rc = tabulate_g_hex8(fK); if (rc) return rc; }
    const char *sm = getenv("TB_SYM_MODE");
    const bool onepass = sm && !strcmp(sm, "onepass"); // two accumulator blocks, one pass (smaller tiles)
    const bool two = fK && fM, seq = !(two && onepass);
    const int nreg = two && !seq ? 2 : 1;
    rc = ensure_patch_fused(m, p, nreg, true);
    if (rc) return rc;
    const PatchPlan *pp = m->patches.get();
    const PatchFusedPlan *pf = p->patch_fused.get();
    if (!pf->d_hdr) { set_error("symmetric patch kernel: a patch count does not fit the packed header"); return TB_ERR_UNSUPPORTED; }
    SymView pv{};
    pv.elem_cell = pp->d_elem_cell; pv.elem_ln = (const uint4 *)pf->d_elem_ln; pv.elem_ssig = pf->d_elem_ssig; pv.ssigtab = (const uint4 *)pf->d_ssigtab;
    pv.row_sdesc = (const uint4 *)pf->d_row_sdesc; pv.mirtab = pf->d_mirtab;
    pv.kcap = pf->max_sym_entries; pv.max_rows = pp->max_rows; pv.max_nodes = pf->max_nodes;
    const size_t lds = (size_t)nreg * pv.kcap * sizeof(double) + (size_t)pv.max_rows * 16 + (size_t)pv.max_nodes * 3 * sizeof(double);
#ifdef TB_ABLATION
    static long long *d_prof = nullptr; // TB_PROF_STAMPS: phase time stamps of every 256th workgroup
    const int nprof = (int)(pp->n_patches >> 8) + 1;
    if (!d_prof && getenv("TB_PROF_STAMPS")) TB_HIP(hipMalloc((void **)&d_prof, (size_t)nprof * 16 * sizeof(long long)));
    if (d_prof) TB_HIP(hipMemsetAsync(d_prof, 0, (size_t)nprof * 16 * sizeof(long long), dev->stream));
    pv.prof = d_prof;
#endif
    const FormArgs aK = fK ? make_args(fK, t) : FormArgs{};
    const FormArgs aM = fM ? make_args(fM, t) : FormArgs{};
    const bool fk = fK && fK->field, fm = fM && fM->field;
    // constant diagonal tensors (isotropic / axis-aligned conductivities) take the variant with the cheaper A·D product
    const bool diag = fK && !fk && aK.D[1] == 0.0 && aK.D[2] == 0.0 && aK.D[5] == 0.0 && aK.D[3] == 0.0 && aK.D[6] == 0.0 && aK.D[7] == 0.0;
    auto launch = [&](auto k) -> int {
        TB_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k, dim3((unsigned)pp->n_patches), dim3(256), lds, dev->stream, aK, aM, pv, (const uint4 *)pf->d_hdr, pf->d_pcoord, d_nzK, d_nzM, dev->d_status);
        return TB_OK;
    };
#define TB_SL(a, b, c, d, e) (seq ? launch(k_patch_hex8_sym<a, b, c, d, e, true>) : launch(k_patch_hex8_sym<a, b, c, d, e, false>))
    if (two) {
        if (fk && fm) rc = TB_SL(true, true, true, true, false);
        else if (fk) rc = TB_SL(true, true, true, false, false);
        else if (fm) rc = diag ? TB_SL(true, true, false, true, true) : TB_SL(true, true, false, true, false);
        else rc = diag ? TB_SL(true, true, false, false, true) : TB_SL(true, true, false, false, false);
    } else if (fK) {
        if (fk) rc = launch(k_patch_hex8_sym<true, false, true, false, false, true>);
        else rc = diag ? launch(k_patch_hex8_sym<true, false, false, false, true, true>) : launch(k_patch_hex8_sym<true, false, false, false, false, true>);
    } else {
        rc = fm ? launch(k_patch_hex8_sym<false, true, false, true, false, true>) : launch(k_patch_hex8_sym<false, true, false, false, false, true>);
    }
#undef TB_SL
    if (rc) return rc;
    TB_HIP(hipGetLastError());
#ifdef TB_ABLATION
    if (d_prof) { // average phase durations (µs; wall clock 100 MHz)
        std::vector<long long> hst((size_t)nprof * 16);
        TB_HIP(hipStreamSynchronize(dev->stream));
        TB_HIP(hipMemcpy(hst.data(), d_prof, hst.size() * sizeof(long long), hipMemcpyDeviceToHost));
        double ph[10] = {0};
        int cnt = 0;
        for (int w = 0; w < nprof; ++w) {
            const long long *a = &hst[(size_t)w * 16];
            if (!a[0] || !a[5]) continue;
            for (int k = 0; k < 9; ++k) ph[k] += a[k + 1] ? (double)(a[k + 1] - a[k]) * 0.01 : 0.0;
            ++cnt;
        }
        if (cnt) fprintf(stderr, "[tbhip] sym phases (us): issue %.2f | barrier %.2f | integrate-1 %.2f | maps+barrier %.2f | write-out-1 %.2f | zero+barriers %.2f | integrate-2 %.2f | maps+barrier %.2f | write-out-2 %.2f (n=%d)\n",
                         ph[0] / cnt, ph[1] / cnt, ph[2] / cnt, ph[3] / cnt, ph[4] / cnt, ph[5] / cnt, ph[6] / cnt, ph[7] / cnt, ph[8] / cnt, cnt);
    }
#endif
    return check_status(dev);
}

} // namespace tb
