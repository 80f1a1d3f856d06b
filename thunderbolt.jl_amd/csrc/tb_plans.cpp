// tb_plans.cpp — host-side assembly plans, built once per mesh / pattern (the analogue of
// setup_operator, src/solver/interface.jl:17-94: colouring, element-assembly maps, patch decomposition).
#include <algorithm>
#include <parallel/algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>

#include "tb_internal.h"

namespace tb {

// dof → contributing element-vector slots (cell*ndpc + local), ordered by cell
void dof_slots(const tb_mesh *m, std::vector<int64_t> &ptr, std::vector<int32_t> &src)
{
    const int64_t n = m->n_cells * m->ndpc;
    ptr.assign(m->ndofs + 1, 0);
    for (int64_t i = 0; i < n; ++i) ptr[m->h_cell_dofs[i] + 1]++;
    for (int64_t d = 0; d < m->ndofs; ++d) ptr[d + 1] += ptr[d];
    std::vector<int64_t> pos(ptr.begin(), ptr.end() - 1);
    src.resize(n);
    for (int64_t i = 0; i < n; ++i) src[pos[m->h_cell_dofs[i]]++] = (int32_t)i;
}

int build_ea_plan(tb_mesh *m)
{
    if (m->n_cells * m->ndpc >= (int64_t)0x7fffffff) { set_error("element-assembly plan: too many slots for int32"); return TB_ERR_UNSUPPORTED; }
    std::vector<int64_t> ptr;
    std::vector<int32_t> src;
    dof_slots(m, ptr, src);
    auto plan = std::make_unique<EAPlan>();
    int rc = upload(m->dev, ptr, &plan->d_ptr);
    if (rc) return rc;
    rc = upload(m->dev, src, &plan->d_src);
    if (rc) return rc;
    TB_HIP(hipMalloc((void **)&plan->d_ea, sizeof(double) * m->n_cells * m->ndpc));
    m->ea = std::move(plan);
    return TB_OK;
}

// scalar Q2 element strategy: the slot table for element matrices stored in tensor order (slot = cell · 27 + i₀ + 3 i₁ + 9 i₂ of the local row)
int ensure_ea_ell_q2t(tb_mesh *m)
{
    int rc = ensure_ea_ell(m);
    if (rc) return rc;
    if (m->ea->d_ell_t) return TB_OK;
    TB_NO_CAPTURE(m->dev);
    if (m->ndpc != 27) { set_error("tensor-order slot table: needs 27 dofs per cell"); return TB_ERR_UNSUPPORTED; }
    static const uint8_t trow[27] = {0, 2, 8, 6, 18, 20, 26, 24, 1, 5, 7, 3, 19, 23, 25, 21, 9, 11, 17, 15, 4, 10, 14, 16, 12, 22, 13}; // Ferrite node → tensor index
    std::vector<int32_t> ell((size_t)m->ndofs * m->ea->ell_w);
    TB_HIP(hipMemcpy(ell.data(), m->ea->d_ell, ell.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < (int64_t)ell.size(); ++k)
        if (ell[k] >= 0) ell[k] = ell[k] / 27 * 27 + trow[ell[k] % 27];
    return upload(m->dev, ell, &m->ea->d_ell_t);
}

// fixed-width form of the slot lists (width = the largest number of cells at a dof, rounded up to a multiple of 8)
int ensure_ea_ell(tb_mesh *m)
{
    if (!m->ea) { int rc = build_ea_plan(m); if (rc) return rc; }
    if (m->ea->d_ell) return TB_OK;
    std::vector<int64_t> ptr;
    std::vector<int32_t> src;
    dof_slots(m, ptr, src);
    int64_t w = 0;
    for (int64_t d = 0; d < m->ndofs; ++d) w = std::max(w, ptr[d + 1] - ptr[d]);
    w = (w + 7) / 8 * 8;
    std::vector<int32_t> ell((size_t)m->ndofs * w, -1);
    for (int64_t d = 0; d < m->ndofs; ++d) std::copy(src.begin() + ptr[d], src.begin() + ptr[d + 1], ell.begin() + d * w);
    int rc = upload(m->dev, ell, &m->ea->d_ell);
    if (rc) return rc;
    m->ea->ell_w = (int)w;
    m->ea->h_done.resize((size_t)m->ndofs);
    int32_t run = 0;
    for (int64_t d = 0; d < m->ndofs; ++d) { // slots of a dof are cell-ordered: the last one is its last cell
        if (ptr[d + 1] > ptr[d]) run = std::max(run, src[ptr[d + 1] - 1] / m->ndpc);
        m->ea->h_done[d] = run;
    }
    return TB_OK;
}

int ensure_aux_stream(tb_device *dev)
{
    if (dev->aux_stream) return TB_OK;
    static const bool hi = [] { const char *e = tune_env("TB_MECH_CHUNK_PRIO"); return e && atoi(e); }();
    int lo_p = 0, hi_p = 0;
    TB_HIP(hipDeviceGetStreamPriorityRange(&lo_p, &hi_p));
    TB_HIP(hipStreamCreateWithPriority(&dev->aux_stream, hipStreamNonBlocking, hi ? hi_p : lo_p));
    for (hipEvent_t &e : dev->aux_ev) TB_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return TB_OK;
}

// Greedy colouring of the cell conflict graph (two cells conflict iff they share a dof): smallest
// colour not used by any neighbour, neighbours found through a per-dof colour bitmask.
int build_color_plan(tb_mesh *m)
{
    std::vector<uint64_t> used(m->ndofs, 0);
    std::vector<int32_t> color(m->n_cells);
    int ncolors = 0;
    for (int64_t c = 0; c < m->n_cells; ++c) {
        const int32_t *d = &m->h_cell_dofs[c * m->ndpc];
        uint64_t mask = 0;
        for (int l = 0; l < m->ndpc; ++l) mask |= used[d[l]];
        const int col = mask == ~0ull ? 64 : __builtin_ctzll(~mask);
        if (col >= 64) { set_error("colouring needs more than 64 colours"); return TB_ERR_UNSUPPORTED; }
        color[c] = col;
        ncolors = std::max(ncolors, col + 1);
        for (int l = 0; l < m->ndpc; ++l) used[d[l]] |= 1ull << col;
    }
    auto plan = std::make_unique<ColorPlan>();
    plan->ncolors = ncolors;
    plan->offsets.assign(ncolors + 1, 0);
    for (int64_t c = 0; c < m->n_cells; ++c) plan->offsets[color[c] + 1]++;
    for (int k = 0; k < ncolors; ++k) plan->offsets[k + 1] += plan->offsets[k];
    std::vector<int64_t> pos(plan->offsets.begin(), plan->offsets.end() - 1);
    std::vector<int32_t> cells(m->n_cells);
    for (int64_t c = 0; c < m->n_cells; ++c) cells[pos[color[c]]++] = (int32_t)c;
    int rc = upload(m->dev, cells, &plan->d_cells);
    if (rc) return rc;
    m->colors = std::move(plan);
    return TB_OK;
}

// the same greedy colouring over a subset of the cells (subdomain forms: one integrator per SubDofHandler in the reference)
int build_color_plan_subset(tb_mesh *m, const std::vector<int32_t> &cells, std::unique_ptr<ColorPlan> &out)
{
    std::vector<uint64_t> used(m->ndofs, 0);
    std::vector<int32_t> color(cells.size());
    int ncolors = 0;
    for (size_t i = 0; i < cells.size(); ++i) {
        const int32_t *d = &m->h_cell_dofs[(int64_t)cells[i] * m->ndpc];
        uint64_t mask = 0;
        for (int l = 0; l < m->ndpc; ++l) mask |= used[d[l]];
        const int col = mask == ~0ull ? 64 : __builtin_ctzll(~mask);
        if (col >= 64) { set_error("colouring needs more than 64 colours"); return TB_ERR_UNSUPPORTED; }
        color[i] = col;
        ncolors = std::max(ncolors, col + 1);
        for (int l = 0; l < m->ndpc; ++l) used[d[l]] |= 1ull << col;
    }
    auto plan = std::make_unique<ColorPlan>();
    plan->ncolors = ncolors;
    plan->offsets.assign(ncolors + 1, 0);
    for (size_t i = 0; i < cells.size(); ++i) plan->offsets[color[i] + 1]++;
    for (int k = 0; k < ncolors; ++k) plan->offsets[k + 1] += plan->offsets[k];
    std::vector<int64_t> pos(plan->offsets.begin(), plan->offsets.end() - 1);
    std::vector<int32_t> grouped(cells.size());
    for (size_t i = 0; i < cells.size(); ++i) grouped[pos[color[i]]++] = cells[i];
    if (!grouped.empty()) { int rc = upload(m->dev, grouped, &plan->d_cells); if (rc) return rc; }
    out = std::move(plan);
    return TB_OK;
}

static inline uint64_t spread21(uint64_t v)
{
    v &= 0x1fffff;
    v = (v | v << 32) & 0x1f00000000ffffull;
    v = (v | v << 16) & 0x1f0000ff0000ffull;
    v = (v | v << 8) & 0x100f00f00f00f00full;
    v = (v | v << 4) & 0x10c30c30c30c30c3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}

// Extent of a hexahedron along the three coordinate axes for the layer count of the bucket construction: per axis the largest component of the
// three face-centre-to-face-centre vectors.  On a distorted structured grid these average to the layer spacing exactly (they telescope along a
// grid line), so extent / mean comes out as the number of layers; the bounding box of a sheared cell is larger than the spacing (216 layers of a
// mesh perturbed by 0.2 h counted as ≈ 200 buckets: tiles then cut through the layers, ragged faces, patches of 270 instances and 158 rows).
static inline void hex_axis_extents(const double *xyz, const int32_t *cn, double (&ext)[3])
{
    static const int lo_hi[3][2][4] = {{{0, 3, 4, 7}, {1, 2, 5, 6}}, {{0, 1, 4, 5}, {3, 2, 7, 6}}, {{0, 1, 2, 3}, {4, 5, 6, 7}}};
    ext[0] = ext[1] = ext[2] = 0.0;
    for (int k = 0; k < 3; ++k) {
        double e[3] = {0, 0, 0};
        for (int a = 0; a < 4; ++a)
            for (int d = 0; d < 3; ++d) e[d] += 0.25 * (xyz[3 * (int64_t)cn[lo_hi[k][1][a]] + d] - xyz[3 * (int64_t)cn[lo_hi[k][0][a]] + d]);
        for (int d = 0; d < 3; ++d) ext[d] = std::max(ext[d], std::fabs(e[d]));
    }
}

// Tile of bucket b (0 ≤ b < R) along one axis, for tiles of at most `tile` layers, split evenly instead of cutting full tiles and leaving a sliver:
//   by_nodes: the R + 1 node layers are dealt to ⌈(R + 1) / tile⌉ tiles in balanced groups and a cell goes with its upper node layer (rows are
//   owned by first touch, so the tile of cell b owns node layer b + 1 and the first tile node layer 0 as well) — no tile owns more than `tile`
//   layers of rows, the domain-boundary tiles included (cutting at multiples of `tile` gave the first tile tile + 1 layers: too many rows for the
//   LDS block, so those patches were split again, and 216 = 43 × 5 + 1 left a one-cell sliver of tiles along two faces);
//   otherwise the R cell layers are dealt to ⌈R / tile⌉ tiles (patches that own cells, not rows).
static inline uint64_t balanced_tile(uint32_t b, int64_t R, int tile, bool by_nodes)
{
    if (by_nodes) { const int64_t nt = (R + 1 + tile - 1) / tile; return (uint64_t)(((int64_t)b + 1) * nt / (R + 1)); }
    const int64_t nt = (R + tile - 1) / tile;
    return (uint64_t)((int64_t)b * nt / R);
}

// Patch decomposition.  Cells are ordered along a Morton curve through their centroids (works for any
// unstructured mesh), cut into runs of `cells_per_patch`; a dof is owned by the first patch (in that
// order) that touches it; a patch's element list = every cell touching one of its rows.
// Recursive bisection of the cell centroids into ⌈nc / target⌉ leaves of equal size (± 1 cell).  Every split cuts the current subset across the axis
// along which it holds the most cell LAYERS (extent ÷ mean cell extent: the halo of a patch counts cells, and anisotropic cells — the ventricle's are
// ≈ 2 : 2 : 1 — would otherwise give leaves twice as long in cells), in proportion to the leaves either side receives.  Tiles of per-axis buckets fill a
// curved thin-walled mesh badly (the idealised ventricle: 132 instances per 256-lane patch, 2.28 per cell); leaves follow the geometry.
static void bisect_cells(const std::vector<double> &cen, const std::vector<double> &cext, int64_t nc, int64_t target, std::vector<int32_t> &leaf)
{
    const int64_t L = (nc + target - 1) / target;
    std::vector<int32_t> ids(nc);
    for (int64_t c = 0; c < nc; ++c) ids[c] = (int32_t)c;
    struct Job { int64_t b, e, l0, nl; };
    std::vector<Job> stack{{0, nc, 0, L}};
    leaf.assign(nc, 0);
    while (!stack.empty()) {
        const Job j = stack.back(); stack.pop_back();
        if (j.nl <= 1 || j.e - j.b <= 1) { for (int64_t k = j.b; k < j.e; ++k) leaf[ids[k]] = (int32_t)j.l0; continue; }
        double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300}, es[3] = {0, 0, 0};
        for (int64_t k = j.b; k < j.e; ++k)
            for (int d = 0; d < 3; ++d) {
                const double v = cen[3 * (size_t)ids[k] + d];
                mn[d] = std::min(mn[d], v); mx[d] = std::max(mx[d], v); es[d] += cext[3 * (size_t)ids[k] + d];
            }
        int ax = 0;
        double best = -1.0;
        for (int d = 0; d < 3; ++d) {
            const double layers = es[d] > 0.0 ? (mx[d] - mn[d]) * (double)(j.e - j.b) / es[d] : 0.0;
            if (layers > best) { best = layers; ax = d; }
        }
        const int64_t nl_left = j.nl / 2, mid = j.b + (j.e - j.b) * nl_left / j.nl;
        std::nth_element(ids.begin() + j.b, ids.begin() + mid, ids.begin() + j.e,
                         [&](int32_t a, int32_t b) { const double va = cen[3 * (size_t)a + ax], vb = cen[3 * (size_t)b + ax]; return va < vb || (va == vb && a < b); });
        stack.push_back({mid, j.e, j.l0 + nl_left, j.nl - nl_left});
        stack.push_back({j.b, mid, j.l0, nl_left});
    }
}

int build_patch_plan(tb_mesh *m, int cells_per_patch)
{
    // Patch shape.  Default: tiles of 7×7×7 bucket cells — 343 owned rows (≈80 KB of LDS accumulators, two
    // workgroups per CU) and 8³ = 512 cell instances = exactly two full 256-thread sweeps.  TB_PATCH_TILE="tx,ty,tz"
    // overrides; TB_PATCH_CELLS=n (or the argument) selects Morton runs of n cells instead.
    int tile[3] = {7, 7, 7};
    const bool default_tile = cells_per_patch == 0 && !getenv("TB_PATCH_TILE");
    bool use_tiles = cells_per_patch <= 0 && !tune_env("TB_PATCH_CELLS");
    if (const char *e = getenv("TB_PATCH_TILE")) {
        int a, b, c;
        if (sscanf(e, "%d,%d,%d", &a, &b, &c) == 3 && a > 0 && b > 0 && c > 0) { tile[0] = a; tile[1] = b; tile[2] = c; use_tiles = cells_per_patch <= 0; }
    }
    if (cells_per_patch < 0 && m->patch_rcb > 0) { // shrink request under the bisection plan: a smaller leaf
        m->patch_rcb = std::max(16, m->patch_rcb * 7 / 8);
        cells_per_patch = 0;
    }
    if (cells_per_patch < 0) { // shrink request from the LDS-fit retry loop: −k → k-th reduction of the tile
        for (int k = 0; k < -cells_per_patch; ++k) { int d = 0; for (int j = 1; j < 3; ++j) if (tile[j] > tile[d]) d = j; if (tile[d] > 1) --tile[d]; }
        cells_per_patch = 0;
    }
    if (use_tiles) cells_per_patch = tile[0] * tile[1] * tile[2];
    else if (cells_per_patch <= 0) {
        const char *e = tune_env("TB_PATCH_CELLS");
        cells_per_patch = e ? atoi(e) : 256;
        if (cells_per_patch <= 0) cells_per_patch = 256;
    }
    const int64_t nc = m->n_cells;
    const int ndpc = m->ndpc, nv = m->nverts;
    PlanTimer timer("build_patch_plan");
    // 1. Morton order over quantile buckets.  Per axis the cells are ranked by centroid coordinate and cut into
    //    R_d equal-count buckets, R_d = extent / mean cell extent: on (mildly distorted) structured grids the
    //    buckets are exactly the cell layers (i,j,k), on unstructured meshes they adapt to the local density.
    std::vector<std::pair<uint64_t, int32_t>> keyed(nc);
    std::vector<uint32_t> bucket((size_t)nc * 3);
    int64_t Rv[3] = {1, 1, 1};
    {
        std::vector<double> cen((size_t)nc * 3), cext((size_t)nc * 3);
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300}, hsum[3] = {0, 0, 0};
#pragma omp parallel for schedule(static) reduction(min : lo[:3]) reduction(max : hi[:3]) reduction(+ : hsum[:3])
        for (int64_t c = 0; c < nc; ++c) {
            double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300}, sum[3] = {0, 0, 0};
            for (int a = 0; a < nv; ++a)
                for (int d = 0; d < 3; ++d) {
                    const double v = m->h_xyz[3 * (int64_t)m->h_conn[c * nv + a] + d];
                    sum[d] += v; mn[d] = std::min(mn[d], v); mx[d] = std::max(mx[d], v);
                }
            double ext[3] = {mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]};
            if (nv == 8) hex_axis_extents(&m->h_xyz[0], &m->h_conn[c * 8], ext);
            for (int d = 0; d < 3; ++d) {
                cen[3 * c + d] = sum[d] / nv;
                cext[3 * c + d] = ext[d];
                lo[d] = std::min(lo[d], mn[d]); hi[d] = std::max(hi[d], mx[d]); hsum[d] += ext[d];
            }
        }
        for (int d = 0; d < 3; ++d) { // (round 6: one axis after the other, every sort on the whole team — three one-thread sorts of 10⁷ pairs took 1.0 s)
            const double hmean = hsum[d] / (double)std::max<int64_t>(nc, 1);
            int64_t R = hmean > 0 ? (int64_t)std::llround((hi[d] - lo[d]) / hmean) : 1;
            R = std::min<int64_t>(std::max<int64_t>(R, 1), 1 << 21);
            Rv[d] = R;
            std::vector<std::pair<double, int32_t>> byc(nc);
#pragma omp parallel for schedule(static)
            for (int64_t c = 0; c < nc; ++c) byc[c] = {cen[3 * c + d], (int32_t)c};
            __gnu_parallel::sort(byc.begin(), byc.end());
#pragma omp parallel for schedule(static)
            for (int64_t r = 0; r < nc; ++r) bucket[3 * (size_t)byc[r].second + d] = (uint32_t)((r * R) / nc);
        }
        // A bucket holds more than one cell where cells overlap in every coordinate — the six tetrahedra of a split hexahedron span the same box.  A
        // tile is then cut only at its boundary, not after tile-volume cells (runs of 343 tetrahedra were flat slabs with four times their own
        // number of halo instances), and is 5×5×5 buckets by default (6·10⁶ tetrahedra: mass 0.25 → 0.19 ms, diffusion 0.36 → 0.25 ms, source
        // 0.26 → 0.16 ms; 7×7×7: 0.28 / 0.32 / 0.14 ms; scripts/bench_tets.py)
        if (use_tiles) {
            const int64_t per_bucket = nc / (Rv[0] * Rv[1] * Rv[2]); // 1 on hexahedral meshes, 6 on their tetrahedral splits
            if (per_bucket >= 2) {
                if (default_tile) tile[0] = tile[1] = tile[2] = 5;
                cells_per_patch = (int)std::min<int64_t>((int64_t)tile[0] * tile[1] * tile[2] * per_bucket, 1 << 20);
            }
        }
        if (m->patch_rcb > 0) { // chosen by ensure_patch_fused when the tile plan's fill is poor (round 5)
            std::vector<int32_t> leaf;
            bisect_cells(cen, cext, nc, m->patch_rcb, leaf);
            use_tiles = true; // a leaf is a tile: a patch ends where the key changes
            cells_per_patch = m->patch_rcb + 1;
#pragma omp parallel for schedule(static)
            for (int64_t c = 0; c < nc; ++c) keyed[c] = {(uint64_t)leaf[c], (int32_t)c};
        } else {
        static const bool legacy_cut = getenv("TB_PATCH_CUT") && !strcmp(getenv("TB_PATCH_CUT"), "full"); // full tiles + sliver (rounds 1–3), for A/B runs
        auto tile_of = [&](uint32_t b, int d) -> uint64_t { return legacy_cut ? (uint64_t)(b / tile[d]) : balanced_tile(b, Rv[d], tile[d], true); };
#pragma omp parallel for schedule(static)
        for (int64_t c = 0; c < nc; ++c) {
            if (use_tiles) { // tile id (z-major), 21 bits per axis
                const uint64_t ti = tile_of(bucket[3 * c], 0), tj = tile_of(bucket[3 * c + 1], 1), tk = tile_of(bucket[3 * c + 2], 2);
                keyed[c] = {(tk << 42) | (tj << 21) | ti, (int32_t)c};
            } else {
                keyed[c] = {spread21(bucket[3 * c]) | spread21(bucket[3 * c + 1]) << 1 | spread21(bucket[3 * c + 2]) << 2, (int32_t)c};
            }
        }
        }
    }
    timer.lap("buckets, keys");
    __gnu_parallel::sort(keyed.begin(), keyed.end());
    timer.lap("key sort");
    // 2. patch boundaries + row ownership by first touch, in Morton order.  A patch closes after `cells_per_patch`
    //    cells or when it would own more than 9/8 of that many rows (domain-boundary patches own the extra
    //    boundary layers), which bounds the LDS accumulator block of every workgroup.
    const int max_rows_cfg = m->patch_rcb > 0 ? m->patch_rcb + m->patch_rcb / 4 // leaves at the surface of the domain own its extra node layers
                                              : use_tiles ? cells_per_patch + 8 : cells_per_patch + cells_per_patch / 8;
    std::vector<int32_t> owner(m->ndofs, -1);
    std::vector<int64_t> pstart(1, 0);
    {
        int rows_in = 0, cells_in = 0;
        for (int64_t k = 0; k < nc; ++k) {
            const int32_t c = keyed[k].second;
            int newrows = 0;
            for (int l = 0; l < ndpc; ++l) newrows += owner[m->h_cell_dofs[(int64_t)c * ndpc + l]] < 0;
            const bool new_tile = use_tiles && k > 0 && keyed[k].first != keyed[k - 1].first;
            if (cells_in == cells_per_patch || new_tile || (cells_in > 0 && rows_in + newrows > max_rows_cfg)) {
                pstart.push_back(k);
                rows_in = 0; cells_in = 0;
            }
            const int32_t pid = (int32_t)pstart.size() - 1;
            for (int l = 0; l < ndpc; ++l) {
                int32_t &o = owner[m->h_cell_dofs[(int64_t)c * ndpc + l]];
                if (o < 0) { o = pid; ++rows_in; }
            }
            ++cells_in;
        }
        pstart.push_back(nc);
    }
    const int64_t np = (int64_t)pstart.size() - 1;
    timer.lap("ownership sweep");
    // Inside a patch, cells (= threads) and hence row slots are ordered lexicographically (x fastest): lanes of a
    // wave then hit consecutive rows of the LDS accumulator block — stride-27 addresses, free of bank conflicts.
    auto lex_less = [&](int32_t a, int32_t b) {
        const uint32_t *A = &bucket[3 * (size_t)a], *B = &bucket[3 * (size_t)b];
        if (A[2] != B[2]) return A[2] < B[2];
        if (A[1] != B[1]) return A[1] < B[1];
        if (A[0] != B[0]) return A[0] < B[0];
        return a < b;
    };
#pragma omp parallel for schedule(static)
    for (int64_t q = 0; q < np; ++q)
        std::sort(keyed.begin() + pstart[q], keyed.begin() + pstart[q + 1],
                  [&](const std::pair<uint64_t, int32_t> &a, const std::pair<uint64_t, int32_t> &b) { return lex_less(a.second, b.second); });
    timer.lap("per-patch sorts");
    // 3. dof → cells
    std::vector<int64_t> sptr;
    std::vector<int32_t> ssrc;
    dof_slots(m, sptr, ssrc);
    timer.lap("dof -> cells");

    auto plan = std::make_unique<PatchPlan>();
    plan->cells_per_patch = cells_per_patch;
    plan->n_patches = np;
    plan->h_elem_ptr.assign(np + 1, 0);
    plan->h_row_ptr.assign(np + 1, 0);
    plan->h_elem_cell.reserve((size_t)(nc * 1.7) + 1024);
    plan->h_row_dof.reserve(m->ndofs);
    plan->h_elem_lrow.reserve((size_t)(nc * 1.7 * ndpc) + 1024);
    std::vector<int32_t> slot_of(m->ndofs, -1);
    std::vector<int32_t> cell_stamp(nc, -1);
    for (int64_t p = 0; p < np; ++p) {
        const int64_t k0 = pstart[p], k1 = pstart[p + 1];
        const size_t row_begin = plan->h_row_dof.size(), elem_begin = plan->h_elem_cell.size();
        // owned rows in first-touch order; own cells first in the element list
        for (int64_t k = k0; k < k1; ++k) {
            const int32_t c = keyed[k].second;
            cell_stamp[c] = (int32_t)p;
            plan->h_elem_cell.push_back(c);
            for (int l = 0; l < ndpc; ++l) {
                const int32_t d = m->h_cell_dofs[(int64_t)c * ndpc + l];
                if (owner[d] == p && slot_of[d] < 0) {
                    slot_of[d] = (int32_t)(plan->h_row_dof.size() - row_begin);
                    plan->h_row_dof.push_back(d);
                }
            }
        }
        // halo cells: every other cell touching an owned row
        const size_t row_end = plan->h_row_dof.size();
        for (size_t r = row_begin; r < row_end; ++r) {
            const int32_t d = plan->h_row_dof[r];
            for (int64_t s = sptr[d]; s < sptr[d + 1]; ++s) {
                const int32_t c = ssrc[s] / ndpc;
                if (cell_stamp[c] != p) { cell_stamp[c] = (int32_t)p; plan->h_elem_cell.push_back(c); }
            }
        }
        const size_t elem_end = plan->h_elem_cell.size();
        std::sort(plan->h_elem_cell.begin() + elem_begin + (k1 - k0), plan->h_elem_cell.begin() + elem_end, lex_less); // halo cells
        if (row_end - row_begin >= 0xFFFF) { set_error("patch owns too many rows (%zu)", row_end - row_begin); return TB_ERR_UNSUPPORTED; }
        for (size_t e = elem_begin; e < elem_end; ++e) {
            const int32_t c = plan->h_elem_cell[e];
            for (int l = 0; l < ndpc; ++l) {
                const int32_t d = m->h_cell_dofs[(int64_t)c * ndpc + l];
                plan->h_elem_lrow.push_back(owner[d] == p ? (uint16_t)slot_of[d] : (uint16_t)0xFFFF);
            }
        }
        for (size_t r = row_begin; r < row_end; ++r) slot_of[plan->h_row_dof[r]] = -1;
        plan->h_elem_ptr[p + 1] = (int64_t)elem_end;
        plan->h_row_ptr[p + 1] = (int64_t)row_end;
        plan->max_elems = std::max<int>(plan->max_elems, (int)(elem_end - elem_begin));
        plan->max_rows = std::max<int>(plan->max_rows, (int)(row_end - row_begin));
    }
    timer.lap("instances of every patch");
    // 256-thread workgroups (two resident per CU at the kernels' register / LDS budget, so one patch's write-out
    // overlaps the other's arithmetic); measured best on MI355X among 64…512
    plan->threads = std::min(256, std::max(64, (plan->max_elems + 63) / 64 * 64));
    if (const char *e = tune_env("TB_PATCH_THREADS")) { const int t = atoi(e); if (t >= 64 && t <= 512 && t % 64 == 0) plan->threads = t; }
    plan->version = m->patches ? m->patches->version + 1 : 1;
    plan->total_elems = (int64_t)plan->h_elem_cell.size();
    plan->total_rows = (int64_t)plan->h_row_dof.size();
    if (plan->total_rows != m->ndofs) { set_error("patch plan: %lld rows owned, %lld dofs", (long long)plan->total_rows, (long long)m->ndofs); return TB_ERR_BAD_ARG; }
    int rc;
    if ((rc = upload(m->dev, plan->h_elem_ptr, &plan->d_elem_ptr))) return rc;
    if ((rc = upload(m->dev, plan->h_row_ptr, &plan->d_row_ptr))) return rc;
    if ((rc = upload(m->dev, plan->h_elem_cell, &plan->d_elem_cell))) return rc;
    if ((rc = upload(m->dev, plan->h_elem_lrow, &plan->d_elem_lrow))) return rc;
    if ((rc = upload(m->dev, plan->h_row_dof, &plan->d_row_dof))) return rc;
    timer.lap("uploads");
    m->patches = std::move(plan);
    return TB_OK;
}

// Matrix extension of the patch plan: LDS offsets of the owned rows and, per element instance, the
// position of every (i,j) coupling inside its row.
int build_patch_mat_plan(tb_pattern *p)
{
    PlanTimer timer("build_patch_mat_plan");
    tb_mesh *m = p->mesh;
    const PatchPlan *pp = m->patches.get();
    const int ndpc = m->ndpc;
    auto plan = std::make_unique<PatchMatPlan>();
    std::vector<RowDesc> row_desc(pp->total_rows);
    int64_t maxlen = 0;
    int64_t max_entries = 0;
    for (int64_t q = 0; q < pp->n_patches; ++q) {
        uint64_t off = 0;
        for (int64_t r = pp->h_row_ptr[q]; r < pp->h_row_ptr[q + 1]; ++r) {
            const int32_t d = pp->h_row_dof[r];
            const int64_t len = p->h_rowptr[d + 1] - p->h_rowptr[d];
            row_desc[r] = RowDesc{p->h_rowptr[d], (uint32_t)off, (uint32_t)len};
            off += (uint64_t)len;
            maxlen = std::max(maxlen, len);
        }
        max_entries = std::max<int64_t>(max_entries, (int64_t)off);
    }
    plan->max_lds_entries = (int)max_entries;
    plan->version = pp->version;
    if (max_entries * 8 + (int64_t)pp->max_rows * 16 > 160 * 1024 || max_entries >= 0xFFFF) {
        set_error("patch plan needs %lld B of LDS per patch (>160 KiB): lower TB_PATCH_TILE / TB_PATCH_CELLS", (long long)(max_entries * 8));
        p->patch_mat = std::move(plan); // keeps the size for the refit loop
        return TB_ERR_UNSUPPORTED;
    }
    const bool wide = maxlen > 255;
    std::vector<uint8_t> cp8;
    std::vector<uint16_t> cp16;
    const size_t n = (size_t)pp->total_elems * ndpc * ndpc;
    if (wide) cp16.assign(n, 0); else cp8.assign(n, 0);
    std::vector<uint16_t> rowoff((size_t)pp->total_elems * ndpc, 0xFFFF);
    std::vector<int32_t> patch_of_elem(pp->total_elems);
    for (int64_t q = 0; q < pp->n_patches; ++q)
        for (int64_t e = pp->h_elem_ptr[q]; e < pp->h_elem_ptr[q + 1]; ++e) patch_of_elem[e] = (int32_t)q;
    bool missing = false;
#pragma omp parallel for schedule(static) reduction(|| : missing)
    for (int64_t e = 0; e < pp->total_elems; ++e) {
        const int32_t c = pp->h_elem_cell[e];
        const int32_t *d = &m->h_cell_dofs[(int64_t)c * ndpc];
        const int64_t r0 = pp->h_row_ptr[patch_of_elem[e]];
        for (int i = 0; i < ndpc; ++i) {
            const uint16_t slot = pp->h_elem_lrow[e * ndpc + i];
            if (slot == 0xFFFF) continue;
            rowoff[e * ndpc + i] = (uint16_t)row_desc[r0 + slot].off;
            const int32_t *b = &p->h_colidx[p->h_rowptr[d[i]]];
            const int32_t *en = &p->h_colidx[p->h_rowptr[d[i] + 1]];
            for (int j = 0; j < ndpc; ++j) {
                const int32_t *it = std::lower_bound(b, en, d[j]);
                if (it == en || *it != d[j]) { missing = true; continue; }
                const size_t k = ((size_t)e * ndpc + i) * ndpc + j;
                if (wide) cp16[k] = (uint16_t)(it - b); else cp8[k] = (uint8_t)(it - b);
            }
        }
    }
    if (missing) { set_error("patch plan: a cell coupling is missing from the CSR pattern"); return TB_ERR_PATTERN; }
    int rc;
    if ((rc = upload(m->dev, row_desc, &plan->d_row_desc))) return rc;
    if ((rc = upload(m->dev, rowoff, &plan->d_elem_rowoff))) return rc;
    if (wide) { if ((rc = upload(m->dev, cp16, &plan->d_colpos16))) return rc; }
    else { if ((rc = upload(m->dev, cp8, &plan->d_colpos8))) return rc; }
    p->patch_mat = std::move(plan);
    return TB_OK;
}

} // namespace tb

namespace tb {

void free_patch_plan(tb_mesh *m)
{
    if (!m->patches) return;
    hipFree(m->patches->d_elem_ptr); hipFree(m->patches->d_row_ptr); hipFree(m->patches->d_elem_cell);
    hipFree(m->patches->d_elem_lrow); hipFree(m->patches->d_row_dof);
    m->patches.reset();
}

void free_patch_mat_plan(tb_pattern *p)
{
    if (!p->patch_mat) return;
    hipFree(p->patch_mat->d_row_desc); hipFree(p->patch_mat->d_elem_rowoff); hipFree(p->patch_mat->d_colpos8); hipFree(p->patch_mat->d_colpos16);
    p->patch_mat.reset();
}

// Build the patch plan and (when a pattern is given) its matrix extension; shrink the tile until one patch's
// LDS block (row accumulators + descriptors) is ≤ 80 KiB, i.e. two workgroups fit a CU's 160 KiB.
int ensure_patch_plans(tb_mesh *m, tb_pattern *p)
{
    if (!m->patches) { int rc = build_patch_plan(m, 0); if (rc) return rc; }
    if (!p) return TB_OK;
    if (p->patch_mat && p->patch_mat->version == m->patches->version) return TB_OK;
    const bool fixed = tune_env("TB_PATCH_CELLS") || getenv("TB_PATCH_TILE");
    for (int attempt = 0; attempt < 12; ++attempt) {
        free_patch_mat_plan(p);
        int rc = build_patch_mat_plan(p);
        const int64_t bytes = p->patch_mat ? (int64_t)p->patch_mat->max_lds_entries * 8 + (int64_t)m->patches->max_rows * 16 : 0;
        if (rc == TB_OK && (fixed || bytes <= 80 * 1024)) return TB_OK;
        if (rc != TB_OK && rc != TB_ERR_UNSUPPORTED) { free_patch_mat_plan(p); return rc; }
        if (fixed) { free_patch_mat_plan(p); return rc; } // no half-built plan is left behind: the next call builds (and reports) again
        const int shrink = m->patches->shrink + 1, version = m->patches->version;
        free_patch_plan(m);
        rc = build_patch_plan(m, -shrink);
        if (rc) return rc;
        m->patches->shrink = shrink;
        if (m->patch_rcb == 0) m->patch_tile_shrink = shrink;
        m->patches->version = version + 1;
    }
    set_error("patch plan: could not fit the LDS budget");
    return TB_ERR_UNSUPPORTED;
}

} // namespace tb

namespace tb {

void free_patch_fused_plan(tb_pattern *p)
{
    if (!p->patch_fused) return;
    PatchFusedPlan *f = p->patch_fused.get();
    hipFree(f->d_node_ptr); hipFree(f->d_pnode); hipFree(f->d_elem_ln); hipFree(f->d_elem_sig); hipFree(f->d_sigtab); hipFree(f->d_row_desc); hipFree(f->d_hdr); hipFree(f->d_pcoord);
    hipFree(f->d_rec);
    p->patch_fused.reset();
}

static inline uint64_t mix64(uint64_t h, uint64_t v)
{
    h ^= v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
    h *= 0xff51afd7ed558ccdull;
    return h ^ (h >> 33);
}

// De-duplication of fixed- or variable-length byte records by content: open addressing on a 64-bit hash, bytes compared on a hit.
struct RecordTable {
    std::vector<int64_t> slot;   // −1 or the offset of the record in `bytes`
    std::vector<uint64_t> hkey;
    std::vector<uint8_t> bytes;  // records back to back, each padded to `stride` (fixed) or stored as [len][payload] (variable)
    int64_t count = 0;
    RecordTable() { slot.assign(1 << 12, -1); hkey.assign(1 << 12, 0); }
    static uint64_t hash(const uint8_t *b, size_t n)
    {
        uint64_t h = 0x243f6a8885a308d3ull ^ n;
        size_t k = 0;
        for (; k + 8 <= n; k += 8) { uint64_t v; memcpy(&v, b + k, 8); h = mix64(h, v); }
        uint64_t v = 0;
        if (k < n) { memcpy(&v, b + k, n - k); h = mix64(h, v); }
        return h;
    }
    void grow()
    {
        std::vector<int64_t> os(slot.size() * 2, -1);
        std::vector<uint64_t> ok(slot.size() * 2, 0);
        const size_t mask = os.size() - 1;
        for (size_t i = 0; i < slot.size(); ++i)
            if (slot[i] >= 0) { size_t j = hkey[i] & mask; while (os[j] >= 0) j = (j + 1) & mask; os[j] = slot[i]; ok[j] = hkey[i]; }
        slot.swap(os); hkey.swap(ok);
    }
    // returns the byte offset of the (possibly new) record; `stored` bytes are appended for a new one (≥ n: padding included by the caller)
    int64_t insert(const uint8_t *b, size_t n, size_t stored)
    {
        if ((size_t)count * 2 >= slot.size()) grow();
        const uint64_t h = hash(b, n);
        const size_t mask = slot.size() - 1;
        size_t j = h & mask;
        while (slot[j] >= 0) {
            if (hkey[j] == h && (size_t)slot[j] + n <= bytes.size() && memcmp(&bytes[slot[j]], b, n) == 0) return slot[j];
            j = (j + 1) & mask;
        }
        const int64_t at = (int64_t)bytes.size();
        bytes.insert(bytes.end(), b, b + n);
        bytes.resize(at + stored, 0);
        slot[j] = at; hkey[j] = h; ++count;
        return at;
    }
};

// Fused-kernel extension of the patch plan (see PatchFusedPlan).  Needs a scalar first-order field whose local dof a sits on
// local vertex a (Ferrite: vertex dofs in vertex order), i.e. a one-to-one node ↔ dof relation; rows of at most 255 entries.
static int build_patch_fused_plan_impl(tb_pattern *p, int64_t *lds_need, int nregions, std::unique_ptr<PatchFusedPlan> &plan)
{
    PlanTimer timer("build_patch_fused_plan_impl");
    tb_mesh *m = p->mesh;
    const PatchPlan *pp = m->patches.get();
    const int ndpc = m->ndpc;
    const bool hex = ndpc == 8 && m->nverts == 8, tet = ndpc == 4 && m->nverts == 4;
    if (!(hex || tet) || m->ncomp != 1) { set_error("fused patch plan: needs a scalar first-order field on hexahedra or tetrahedra"); return TB_ERR_UNSUPPORTED; }
    const int NV = ndpc, NS = NV * NV; // nodes per cell, bytes per position signature
    plan->version = pp->version;
    // 1. row descriptors and the LDS need
    std::vector<RowDesc> row_desc(pp->total_rows);
    int64_t maxlen = 0, max_entries = 0;
    for (int64_t q = 0; q < pp->n_patches; ++q) {
        uint64_t off = 0;
        for (int64_t r = pp->h_row_ptr[q]; r < pp->h_row_ptr[q + 1]; ++r) {
            const int32_t d = pp->h_row_dof[r];
            const int64_t len = p->h_rowptr[d + 1] - p->h_rowptr[d];
            row_desc[r] = RowDesc{p->h_rowptr[d], (uint32_t)off, (uint32_t)len};
            off += (uint64_t)len;
            maxlen = std::max(maxlen, len);
        }
        max_entries = std::max<int64_t>(max_entries, (int64_t)off);
    }
    if (maxlen > 255) { set_error("fused patch plan: a row has %lld entries (> 255)", (long long)maxlen); return TB_ERR_UNSUPPORTED; }
    plan->max_lds_entries = (int)((max_entries + 1) & ~(int64_t)1);
    // 2. per-patch node lists and patch-local node indices of every instance
    std::vector<int64_t> node_ptr(pp->n_patches + 1, 0);
    std::vector<int32_t> pnode;
    pnode.reserve((size_t)(m->n_nodes * 2.2) + 1024);
    std::vector<uint16_t> ln((size_t)pp->total_elems * NV);
    std::vector<int32_t> local_of(m->n_nodes, -1);
    bool bad = false;
    int max_nodes = 0;
    for (int64_t q = 0; q < pp->n_patches; ++q) {
        const int64_t r0 = pp->h_row_ptr[q], nrows = pp->h_row_ptr[q + 1] - r0;
        const size_t base = pnode.size();
        pnode.resize(base + nrows, -1);
        for (int64_t e = pp->h_elem_ptr[q]; e < pp->h_elem_ptr[q + 1]; ++e) { // owned slots first
            const int32_t c = pp->h_elem_cell[e];
            for (int a = 0; a < NV; ++a) {
                const uint16_t slot = pp->h_elem_lrow[e * NV + a];
                if (slot == 0xFFFF) continue;
                const int32_t node = m->h_conn[(int64_t)c * NV + a];
                int32_t &dst = pnode[base + slot];
                if (dst < 0) { dst = node; local_of[node] = slot; }
                else if (dst != node) bad = true;
            }
        }
        for (int64_t e = pp->h_elem_ptr[q]; e < pp->h_elem_ptr[q + 1]; ++e) {
            const int32_t c = pp->h_elem_cell[e];
            for (int a = 0; a < NV; ++a) {
                const int32_t node = m->h_conn[(int64_t)c * NV + a];
                if (local_of[node] < 0) { local_of[node] = (int32_t)(pnode.size() - base); pnode.push_back(node); }
                else if (pp->h_elem_lrow[e * NV + a] == 0xFFFF && local_of[node] < nrows) bad = true; // an owned node reached through a dof the patch does not own
                ln[(size_t)e * NV + a] = (uint16_t)local_of[node];
            }
        }
        const size_t nn = pnode.size() - base;
        if (nn >= 0xFFFF) { set_error("fused patch plan: a patch touches %zu nodes", nn); return TB_ERR_UNSUPPORTED; }
        max_nodes = std::max(max_nodes, (int)nn);
        for (size_t k = base; k < pnode.size(); ++k) if (pnode[k] >= 0) local_of[pnode[k]] = -1; else bad = true;
        node_ptr[q + 1] = (int64_t)pnode.size();
    }
    if (bad) { set_error("fused patch plan: dofs and vertices of the field are not in one-to-one correspondence"); return TB_ERR_UNSUPPORTED; }
    plan->max_nodes = max_nodes;
    const int64_t acc_entries = plan->max_lds_entries;
    *lds_need = (int64_t)nregions * acc_entries * 8 + (int64_t)pp->max_rows * 16 + (int64_t)max_nodes * 24; // + coordinates of the patch's nodes
    if (*lds_need > 80 * 1024 || acc_entries >= 0x7fff) { plan->d_row_desc = nullptr; return TB_ERR_NOMEM; } // caller shrinks the tile and retries (TB_ERR_NOMEM is this function's private "too big")
    // 3. signatures: position of column dof(j) inside row dof(i) for every pair of a cell, de-duplicated
    const int64_t nc = m->n_cells;
    std::vector<uint8_t> sig((size_t)nc * NS);
    std::vector<uint64_t> hash(nc);
    bool missing = false;
#pragma omp parallel for schedule(static) reduction(|| : missing)
    for (int64_t c = 0; c < nc; ++c) {
        const int32_t *d = &m->h_cell_dofs[c * NV];
        uint8_t *sg = &sig[(size_t)c * NS];
        for (int i = 0; i < NV; ++i) {
            const int32_t *b = &p->h_colidx[p->h_rowptr[d[i]]];
            const int32_t *en = &p->h_colidx[p->h_rowptr[d[i] + 1]];
            for (int j = 0; j < NV; ++j) {
                const int32_t *it = std::lower_bound(b, en, d[j]);
                if (it == en || *it != d[j]) { missing = true; sg[i * NV + j] = 0; continue; }
                sg[i * NV + j] = (uint8_t)(it - b);
            }
        }
        uint64_t h = 0x243f6a8885a308d3ull;
        for (int k = 0; k < NS / 8; ++k) { uint64_t v; memcpy(&v, sg + 8 * k, 8); h = mix64(h, v); }
        hash[c] = h;
    }
    if (missing) { set_error("patch plan: a cell coupling is missing from the CSR pattern"); return TB_ERR_PATTERN; }
    std::vector<int32_t> order(nc);
    std::iota(order.begin(), order.end(), 0);
    __gnu_parallel::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
        if (hash[a] != hash[b]) return hash[a] < hash[b];
        const int r = memcmp(&sig[(size_t)a * NS], &sig[(size_t)b * NS], NS);
        return r != 0 ? r < 0 : a < b;
    });
    std::vector<uint32_t> cell_sig(nc);
    std::vector<uint8_t> sigtab;
    int64_t nsig = 0;
    for (int64_t k = 0; k < nc; ++k) {
        const int32_t c = order[k];
        if (k == 0 || memcmp(&sig[(size_t)c * NS], &sig[(size_t)order[k - 1] * NS], NS) != 0) {
            sigtab.insert(sigtab.end(), &sig[(size_t)c * NS], &sig[(size_t)c * NS] + NS);
            ++nsig;
        }
        cell_sig[c] = (uint32_t)(nsig - 1);
    }
    plan->nsig = nsig;
    if (getenv("TB_PLAN_VERBOSE"))
        fprintf(stderr, "[tbhip] fused patch plan: %lld patches, %lld instances (%.3f per cell), max instances/rows/nodes per patch %d/%d/%d, %lld signatures, LDS %lld B for %d block(s)\n",
                (long long)pp->n_patches, (long long)pp->total_elems, (double)pp->total_elems / (double)std::max<int64_t>(nc, 1), pp->max_elems, pp->max_rows, max_nodes,
                (long long)nsig, (long long)*lds_need, nregions);
    std::vector<uint32_t> elem_sig(pp->total_elems);
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < pp->total_elems; ++e) elem_sig[e] = cell_sig[pp->h_elem_cell[e]];
    int rc;
    if ((rc = upload(m->dev, node_ptr, &plan->d_node_ptr))) return rc;
    if ((rc = upload(m->dev, pnode, &plan->d_pnode))) return rc;
    if ((rc = upload(m->dev, ln, &plan->d_elem_ln))) return rc;
    if ((rc = upload(m->dev, elem_sig, &plan->d_elem_sig))) return rc;
    if ((rc = upload(m->dev, sigtab, &plan->d_sigtab))) return rc;
    if ((rc = upload(m->dev, row_desc, &plan->d_row_desc))) return rc;
    if (pp->max_rows < 1024 && max_nodes < 2048 && pp->max_elems < 2048 && pp->total_elems < (int64_t)0x7fffffff && (int64_t)pnode.size() < (int64_t)0x7fffffff) {
        // inputs of the staged kernels: packed headers, pre-gathered coordinates
        std::vector<uint32_t> hdr((size_t)pp->n_patches * 4);
        for (int64_t q = 0; q < pp->n_patches; ++q) {
            hdr[4 * q] = (uint32_t)pp->h_elem_ptr[q]; hdr[4 * q + 1] = (uint32_t)pp->h_row_ptr[q]; hdr[4 * q + 2] = (uint32_t)node_ptr[q];
            const uint32_t ne = (uint32_t)(pp->h_elem_ptr[q + 1] - pp->h_elem_ptr[q]), nr = (uint32_t)(pp->h_row_ptr[q + 1] - pp->h_row_ptr[q]),
                           nn = (uint32_t)(node_ptr[q + 1] - node_ptr[q]);
            hdr[4 * q + 3] = nr | nn << 10 | ne << 21;
        }
        std::vector<double> pcoord(pnode.size() * 3);
#pragma omp parallel for schedule(static)
        for (int64_t k = 0; k < (int64_t)pnode.size(); ++k)
            for (int d = 0; d < 3; ++d) pcoord[3 * k + d] = m->h_xyz[3 * (int64_t)pnode[k] + d];
        if ((rc = upload(m->dev, hdr, &plan->d_hdr))) return rc;
        if ((rc = upload(m->dev, pcoord, &plan->d_pcoord))) return rc;
    }
    return TB_OK;
}

// One-trip records (PatchFusedPlan::d_rec): the arrays of the staged kernel copied patch by patch into fixed-stride records, read back from the device
// copies the plan already holds (the host vectors of the builder are gone by now; this runs once per pattern).
static int ensure_patch_records_impl(tb_pattern *p)
{
    PatchFusedPlan *f = p->patch_fused.get();
    const PatchPlan *pp = p->mesh->patches.get();
    if (!f || !pp || !f->d_hdr) return TB_ERR_UNSUPPORTED;
    if (f->d_rec) return TB_OK;
    TB_NO_CAPTURE(p->mesh->dev);
    PlanTimer timer("ensure_patch_records_impl");
    if (f->rec_stride < 0 || pp->max_elems > 1024 || pp->max_rows > 256) { f->rec_stride = -1; return TB_ERR_UNSUPPORTED; }
    const int64_t np = pp->n_patches;
    const int rm = (pp->max_rows + 3) & ~3, nm = (f->max_nodes + 3) & ~3, nem = std::max(256, (pp->max_elems + 63) & ~63);
    const size_t o_ln = 16, o_sig = o_ln + (size_t)nem * 16, o_desc = o_sig + (size_t)nem * 4, o_xyz = o_desc + (size_t)rm * 16, raw = o_xyz + (size_t)nm * 24;
    const size_t stride = (raw + 127) & ~(size_t)127;
    if (stride * (size_t)np > ((size_t)8 << 30)) { f->rec_stride = -1; return TB_ERR_UNSUPPORTED; } // > 8 GiB of records: keep the compact arrays
    std::vector<uint32_t> hdr((size_t)np * 4);
    std::vector<uint16_t> ln((size_t)pp->total_elems * 8);
    std::vector<uint32_t> sig((size_t)pp->total_elems);
    std::vector<RowDesc> desc((size_t)pp->total_rows);
    TB_HIP(hipMemcpy(hdr.data(), f->d_hdr, hdr.size() * 4, hipMemcpyDeviceToHost));
    TB_HIP(hipMemcpy(ln.data(), f->d_elem_ln, ln.size() * 2, hipMemcpyDeviceToHost));
    TB_HIP(hipMemcpy(sig.data(), f->d_elem_sig, sig.size() * 4, hipMemcpyDeviceToHost));
    TB_HIP(hipMemcpy(desc.data(), f->d_row_desc, desc.size() * sizeof(RowDesc), hipMemcpyDeviceToHost));
    int64_t total_nodes = 0;
    for (int64_t q = 0; q < np; ++q) total_nodes = std::max<int64_t>(total_nodes, (int64_t)hdr[4 * q + 2] + ((hdr[4 * q + 3] >> 10) & 0x7ff));
    std::vector<double> xyz((size_t)total_nodes * 3);
    TB_HIP(hipMemcpy(xyz.data(), f->d_pcoord, xyz.size() * 8, hipMemcpyDeviceToHost));
    std::vector<uint8_t> rec(stride * (size_t)np, 0);
#pragma omp parallel for schedule(static)
    for (int64_t q = 0; q < np; ++q) {
        uint8_t *r = &rec[stride * (size_t)q];
        const uint32_t e0 = hdr[4 * q], r0 = hdr[4 * q + 1], n0 = hdr[4 * q + 2], w = hdr[4 * q + 3];
        const uint32_t nr = w & 0x3ff, nn = (w >> 10) & 0x7ff, ne = w >> 21;
        const uint32_t h4[4] = {w, e0, 0, 0}; // e0: first instance of the patch in the plan's instance arrays (the kernel names the offending cell through it)
        memcpy(r, h4, 16);
        memcpy(r + o_ln, &ln[(size_t)e0 * 8], (size_t)ne * 16);
        memcpy(r + o_sig, &sig[e0], (size_t)ne * 4);
        memcpy(r + o_desc, &desc[r0], (size_t)nr * 16);
        memcpy(r + o_xyz, &xyz[(size_t)n0 * 3], (size_t)nn * 24);
    }
    TB_HIP(hipMalloc((void **)&f->d_rec, rec.size()));
    TB_HIP(hipMemcpy(f->d_rec, rec.data(), rec.size(), hipMemcpyHostToDevice));
    f->rec_stride = (int)stride; f->rec_rm = rm; f->rec_nm = nm; f->rec_ne = nem;
    if (getenv("TB_PLAN_VERBOSE"))
        fprintf(stderr, "[tbhip] one-trip patch records: %lld patches x %zu B (rows <= %d, nodes <= %d), %.2f GB\n", (long long)np, stride, rm, nm, (double)rec.size() / 1e9);
    return TB_OK;
}

// A failure on the way (the records are ≈ 1 GB at 216³: the allocation can fail) is latched — rec_stride = −1, the sticky HIP error cleared — so that the
// following assemblies go straight to the staged kernel instead of repeating four device-to-host copies and a multi-GB host build per call.
int ensure_patch_records(tb_pattern *p)
{
    const int rc = ensure_patch_records_impl(p);
    if (rc != TB_OK && rc != TB_ERR_UNSUPPORTED) {
        PatchFusedPlan *f = p->patch_fused.get();
        if (f) {
            if (f->d_rec) { (void)hipFree(f->d_rec); f->d_rec = nullptr; }
            f->rec_stride = -1;
        }
        (void)hipGetLastError();
        return TB_ERR_UNSUPPORTED;
    }
    return rc;
}

// The plan object is handed to the pattern on every path, so that a failed upload half way leaves nothing behind: the caller's
// free_patch_fused_plan releases whatever was allocated.  TB_ERR_NOMEM from the builder means "does not fit the LDS budget" (size-only plan).
static int build_patch_fused_plan(tb_pattern *p, int64_t *lds_need, int nregions)
{
    auto plan = std::make_unique<PatchFusedPlan>();
    int rc = build_patch_fused_plan_impl(p, lds_need, nregions, plan);
    p->patch_fused = std::move(plan);
    if (rc == TB_ERR_NOMEM) return TB_ERR_UNSUPPORTED; // too big: the caller recognises it by the plan left behind without device arrays
    if (rc != TB_OK) { free_patch_fused_plan(p); return rc; }
    return TB_OK;
}

// Build (or refit) the mesh's patch plan and the pattern's fused extension so that `nregions` blocks of row accumulators, the
// row descriptors and the node list of any patch fit 80 KiB of LDS — two workgroups per CU.
int ensure_patch_fused(tb_mesh *m, tb_pattern *p, int nregions)
{
    if (!m->patches) {
        // Start from the largest tile whose full patch fits the budget by the pattern's longest row, instead of from 7×7×7 with one whole plan build per
        // reduction (216³: six builds of 2.5 s each, 15 of the 21 s of the first assembly — TB_PLAN_VERBOSE=1 prints the stages).  The estimate is the need
        // of a patch that owns every row of its tile; a plan that still does not fit is reduced further by the loop below, as before.
        int k0 = 0;
        const bool fixed0 = tune_env("TB_PATCH_CELLS") || getenv("TB_PATCH_TILE");
        if (!fixed0 && m->patch_rcb == 0 && m->patch_tile_shrink == 0 && m->nverts == 8 && m->ndpc == 8) {
            if (!p->max_row_len) for (int64_t r = 0; r < p->n_rows; ++r) p->max_row_len = std::max<int64_t>(p->max_row_len, p->h_rowptr[r + 1] - p->h_rowptr[r]);
            int t[3] = {7, 7, 7};
            auto fits = [&]() {
                const int64_t rows = (int64_t)t[0] * t[1] * t[2], nodes = (int64_t)(t[0] + 2) * (t[1] + 2) * (t[2] + 2);
                return (int64_t)nregions * rows * p->max_row_len * 8 + rows * 16 + nodes * 24 <= 80 * 1024;
            };
            while (!fits() && (t[0] > 1 || t[1] > 1 || t[2] > 1)) { // the same reduction build_patch_plan applies: the largest extent, first of equals
                int d = 0;
                for (int j = 1; j < 3; ++j) if (t[j] > t[d]) d = j;
                --t[d];
                ++k0;
            }
        }
        int rc = build_patch_plan(m, k0 ? -k0 : 0);
        if (rc) return rc;
        if (k0) { m->patches->shrink = k0; m->patch_tile_shrink = k0; }
    }
    auto need = [&](const PatchFusedPlan *f) {
        return (int64_t)nregions * f->max_lds_entries * 8 + (int64_t)m->patches->max_rows * 16 + (int64_t)f->max_nodes * 24;
    };
    if (p->patch_fused && p->patch_fused->version == m->patches->version && p->patch_fused->d_row_desc &&
        need(p->patch_fused.get()) <= 80 * 1024)
        return TB_OK;
    const bool fixed = tune_env("TB_PATCH_CELLS") || getenv("TB_PATCH_TILE");
    for (int attempt = 0; attempt < 16; ++attempt) {
        free_patch_fused_plan(p);
        int64_t bytes = 0;
        int rc = build_patch_fused_plan(p, &bytes, nregions);
        if (rc == TB_OK) {
            // fill of the 256-lane sweeps: tiles of per-axis buckets are full on box-like meshes (236 of 256 at 216³) and half empty on curved thin-walled
            // ones; there the cells are bisected into equal leaves instead (build_patch_plan, patch_rcb) if that needs clearly fewer patches
            const PatchPlan *pp = m->patches.get();
            const double fill = (double)pp->total_elems / ((double)pp->n_patches * 256.0);
            if (!fixed && m->patch_rcb == 0 && !m->patch_rcb_tried && fill < 0.75 && m->nverts == 8 && m->n_cells >= 4096) {
                m->patch_rcb_tried = true;
                const int64_t np_tiles = pp->n_patches, inst_tiles = pp->total_elems;
                const int version = pp->version;
                m->patch_rcb = 128; // ≈ the own cells of a full 5×5×6-node tile; shrunk by the LDS-fit loop like a tile
                const int rcb_max_inst = tune_env("TB_RCB_MAX_INST") ? atoi(tune_env("TB_RCB_MAX_INST")) : 384; // largest patch (instances) a leaf size may produce before it is shrunk
                free_patch_fused_plan(p);
                free_patch_plan(m);
                rc = build_patch_plan(m, 0);
                if (rc) return rc;
                m->patches->version = version + 1;
                int rc2 = TB_ERR_UNSUPPORTED;
                for (int a2 = 0; a2 < 12; ++a2) {
                    rc2 = build_patch_fused_plan(p, &bytes, nregions);
                    bool big = rc2 == TB_ERR_UNSUPPORTED && p->patch_fused; // does not fit the LDS budget: smaller leaves
                    if (rc2 == TB_OK && m->patches->max_elems > rcb_max_inst && m->patch_rcb > 16) { big = true; rc2 = TB_ERR_UNSUPPORTED; } // a second sweep for many patches: smaller leaves too
                    if (rc2 == TB_OK) break;
                    free_patch_fused_plan(p);
                    if (!big) return rc2;
                    const int v2 = m->patches->version;
                    free_patch_plan(m);
                    rc = build_patch_plan(m, -1);
                    if (rc) return rc;
                    m->patches->version = v2 + 1;
                }
                const bool better = rc2 == TB_OK && m->patches->n_patches * 10 < np_tiles * 9;
                if (getenv("TB_PLAN_VERBOSE"))
                    fprintf(stderr, "[tbhip] patch plan: tiles %lld patches / %lld instances (fill %.2f), bisection (leaf %d) %lld patches / %lld instances -> %s\n", (long long)np_tiles,
                            (long long)inst_tiles, fill, m->patch_rcb, rc2 == TB_OK ? (long long)m->patches->n_patches : -1LL, rc2 == TB_OK ? (long long)m->patches->total_elems : -1LL,
                            better ? "bisection" : "tiles");
                if (better) return TB_OK;
                // back to the tiles
                const int v3 = m->patches->version;
                m->patch_rcb = 0;
                free_patch_fused_plan(p);
                free_patch_plan(m);
                rc = build_patch_plan(m, m->patch_tile_shrink ? -m->patch_tile_shrink : 0);
                if (rc) return rc;
                m->patches->shrink = m->patch_tile_shrink;
                m->patches->version = v3 + 1;
                continue;
            }
            return TB_OK;
        }
        const bool too_big = rc == TB_ERR_UNSUPPORTED && p->patch_fused; // the builder leaves the size-only plan behind in that case
        free_patch_fused_plan(p);
        if (!too_big) return rc;
        if (fixed) { set_error("patch plan needs %lld B of LDS per patch (> 80 KiB): lower TB_PATCH_TILE / TB_PATCH_CELLS", (long long)bytes); return TB_ERR_UNSUPPORTED; }
        const int shrink = m->patches->shrink + 1, version = m->patches->version;
        free_patch_plan(m);
        rc = build_patch_plan(m, -shrink);
        if (rc) return rc;
        m->patches->shrink = shrink;
        m->patches->version = version + 1;
    }
    set_error("patch plan: could not fit the LDS budget");
    return TB_ERR_UNSUPPORTED;
}

} // namespace tb

namespace tb {

void free_vec_patch_plans(tb_mesh *m)
{
    for (auto &v : m->vpatches) {
        if (!v) continue;
        hipFree(v->d_hdr); hipFree(v->d_elem_ln); hipFree(v->d_elem_cell); hipFree(v->d_pcoord); hipFree(v->d_pdof);
        v.reset();
    }
}

// Tiles of 8×8×8 bucket cells (TB_VPATCH_TILE="tx,ty,tz" overrides), cells ordered lexicographically inside a tile — the same bucket construction
// as build_patch_plan (quantile buckets per axis: exact layers on structured boxes, density-adaptive on unstructured meshes).
int ensure_vec_patch_plan(tb_mesh *m, bool halo)
{
    if (m->vpatches[halo]) return TB_OK;
    if (m->ndpc != 8 || m->nverts != 8 || m->ncomp != 1) { set_error("vector patch plan: needs a scalar trilinear hexahedron field"); return TB_ERR_UNSUPPORTED; }
    PlanTimer timer("ensure_vec_patch_plan");
    int tile[3] = {8, 8, 8};
    if (const char *e = tune_env("TB_VPATCH_TILE")) {
        int a, b, c;
        if (sscanf(e, "%d,%d,%d", &a, &b, &c) == 3 && a > 0 && b > 0 && c > 0) { tile[0] = a; tile[1] = b; tile[2] = c; }
    }
    const int64_t nc = m->n_cells;
    std::vector<uint32_t> bucket((size_t)nc * 3);
    std::vector<std::pair<uint64_t, int32_t>> keyed(nc);
    int64_t Rv[3] = {1, 1, 1};
    {
        std::vector<double> cen((size_t)nc * 3), cext((size_t)nc * 3);
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300}, hsum[3] = {0, 0, 0};
#pragma omp parallel for schedule(static) reduction(min : lo[:3]) reduction(max : hi[:3]) reduction(+ : hsum[:3])
        for (int64_t c = 0; c < nc; ++c) {
            double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300}, sum[3] = {0, 0, 0};
            for (int a = 0; a < 8; ++a)
                for (int d = 0; d < 3; ++d) {
                    const double v = m->h_xyz[3 * (int64_t)m->h_conn[c * 8 + a] + d];
                    sum[d] += v; mn[d] = std::min(mn[d], v); mx[d] = std::max(mx[d], v);
                }
            double ext[3];
            hex_axis_extents(&m->h_xyz[0], &m->h_conn[c * 8], ext);
            for (int d = 0; d < 3; ++d) {
                cen[3 * c + d] = sum[d] / 8;
                cext[3 * c + d] = ext[d];
                lo[d] = std::min(lo[d], mn[d]); hi[d] = std::max(hi[d], mx[d]); hsum[d] += ext[d];
            }
        }
        for (int d = 0; d < 3; ++d) { // (round 6: one axis after the other, every sort on the whole team — three one-thread sorts of 10⁷ pairs took 1.0 s)
            const double hmean = hsum[d] / (double)std::max<int64_t>(nc, 1);
            int64_t R = hmean > 0 ? (int64_t)std::llround((hi[d] - lo[d]) / hmean) : 1;
            R = std::min<int64_t>(std::max<int64_t>(R, 1), 1 << 21);
            Rv[d] = R;
            std::vector<std::pair<double, int32_t>> byc(nc);
#pragma omp parallel for schedule(static)
            for (int64_t c = 0; c < nc; ++c) byc[c] = {cen[3 * c + d], (int32_t)c};
            __gnu_parallel::sort(byc.begin(), byc.end());
#pragma omp parallel for schedule(static)
            for (int64_t r = 0; r < nc; ++r) bucket[3 * (size_t)byc[r].second + d] = (uint32_t)((r * R) / nc);
        }
        static const bool legacy_cut = getenv("TB_PATCH_CUT") && !strcmp(getenv("TB_PATCH_CUT"), "full");
#pragma omp parallel for schedule(static)
        for (int64_t c = 0; c < nc; ++c) {
            uint64_t t3[3];
            for (int d = 0; d < 3; ++d) t3[d] = legacy_cut ? (uint64_t)(bucket[3 * c + d] / tile[d]) : balanced_tile(bucket[3 * c + d], Rv[d], tile[d], halo);
            keyed[c] = {(t3[2] << 42) | (t3[1] << 21) | t3[0], (int32_t)c};
        }
        // Poorly filled tiles (a curved thin wall: most tiles of the bucket grid hold a fraction of their volume, and the halo of a fragment is as large
        // as a full tile's): equal leaves of the recursive bisection instead (bisect_cells; the matrix plan's rule, decided here on this plan's own fill
        // so that the outcome does not depend on which plan of the mesh is built first).
        {
            std::vector<uint64_t> keys(nc);
            for (int64_t c = 0; c < nc; ++c) keys[c] = keyed[c].first;
            __gnu_parallel::sort(keys.begin(), keys.end());
            const int64_t ntiles = std::unique(keys.begin(), keys.end()) - keys.begin();
            const int64_t vol = (int64_t)tile[0] * tile[1] * tile[2];
            const double fill = ntiles > 0 ? (double)nc / ((double)ntiles * (double)vol) : 1.0;
            const char *e = tune_env("TB_PATCH_RCB");
            if (nc >= 4096 && fill < 0.75 && !(e && !strcmp(e, "0"))) {
                std::vector<int32_t> leaf;
                bisect_cells(cen, cext, nc, vol * 7 / 8, leaf);
#pragma omp parallel for schedule(static)
                for (int64_t c = 0; c < nc; ++c) keyed[c] = {(uint64_t)leaf[c], (int32_t)c};
                if (getenv("TB_PLAN_VERBOSE")) fprintf(stderr, "[tbhip] vector patch plan: tile fill %.2f -> bisection, %ld leaves of <= %ld cells\n", fill, (long)((nc + vol * 7 / 8 - 1) / (vol * 7 / 8)), (long)(vol * 7 / 8));
            }
        }
    }
    auto lex_less = [&](int32_t a, int32_t b) {
        const uint32_t *A = &bucket[3 * (size_t)a], *B = &bucket[3 * (size_t)b];
        if (A[2] != B[2]) return A[2] < B[2];
        if (A[1] != B[1]) return A[1] < B[1];
        if (A[0] != B[0]) return A[0] < B[0];
        return a < b;
    };
    __gnu_parallel::sort(keyed.begin(), keyed.end(), [&](const std::pair<uint64_t, int32_t> &a, const std::pair<uint64_t, int32_t> &b) {
        return a.first != b.first ? a.first < b.first : lex_less(a.second, b.second);
    });
    // patches: one per tile, cut further so that no patch exceeds 1000 cells (16-bit local node indices, LDS)
    const int cap = std::min(1000, tile[0] * tile[1] * tile[2]);
    std::vector<int64_t> pstart(1, 0);
    for (int64_t k = 1; k < nc; ++k)
        if (keyed[k].first != keyed[k - 1].first || k - pstart.back() >= cap) pstart.push_back(k);
    if (nc > 0) pstart.push_back(nc);
    const int64_t np = (int64_t)pstart.size() - 1;
    // first-touch ownership (halo variant)
    std::vector<int32_t> owner;
    std::vector<int64_t> sptr;
    std::vector<int32_t> ssrc;
    if (halo) {
        owner.assign(m->ndofs, -1);
        for (int64_t q = 0; q < np; ++q)
            for (int64_t k = pstart[q]; k < pstart[q + 1]; ++k)
                for (int l = 0; l < 8; ++l) {
                    int32_t &o = owner[m->h_cell_dofs[(int64_t)keyed[k].second * 8 + l]];
                    if (o < 0) o = (int32_t)q;
                }
        dof_slots(m, sptr, ssrc);
    }
    auto plan = std::make_unique<VecPatchPlan>();
    plan->halo = halo;
    plan->n_patches = np;
    std::vector<uint32_t> hdr((size_t)np * 4);
    std::vector<uint16_t> ln;
    std::vector<int32_t> ecell, pdof, pnode;
    ln.reserve((size_t)nc * 8 * (halo ? 3 : 2) / 2);
    ecell.reserve((size_t)nc * (halo ? 3 : 2) / 2);
    std::vector<int32_t> local_of(m->n_nodes, -1), cell_stamp(halo ? nc : 0, -1);
    std::vector<int32_t> cells;
    bool bad = false;
    for (int64_t q = 0; q < np; ++q) {
        cells.clear();
        for (int64_t k = pstart[q]; k < pstart[q + 1]; ++k) cells.push_back(keyed[k].second);
        const size_t nbase = pnode.size();
        if (halo) { // owned nodes first (slot order = first touch inside the patch), then the halo cells
            for (int32_t c : cells) {
                cell_stamp[c] = (int32_t)q;
                for (int a = 0; a < 8; ++a) {
                    const int32_t d = m->h_cell_dofs[(int64_t)c * 8 + a], node = m->h_conn[(int64_t)c * 8 + a];
                    if (owner[d] == q && local_of[node] < 0) { local_of[node] = (int32_t)(pnode.size() - nbase); pnode.push_back(node); pdof.push_back(d); }
                }
            }
            const size_t nown = pnode.size() - nbase;
            const size_t nown_cells = cells.size();
            for (size_t r = 0; r < nown; ++r) {
                const int32_t d = pdof[nbase + r];
                for (int64_t s2 = sptr[d]; s2 < sptr[d + 1]; ++s2) {
                    const int32_t c = ssrc[s2] / 8;
                    if (cell_stamp[c] != q) { cell_stamp[c] = (int32_t)q; cells.push_back(c); }
                }
            }
            std::sort(cells.begin() + nown_cells, cells.end(), lex_less);
            hdr[4 * q + 2] = (uint32_t)nown;
        }
        for (int32_t c : cells)
            for (int a = 0; a < 8; ++a) {
                const int32_t d = m->h_cell_dofs[(int64_t)c * 8 + a], node = m->h_conn[(int64_t)c * 8 + a];
                if (local_of[node] < 0) { local_of[node] = (int32_t)(pnode.size() - nbase); pnode.push_back(node); pdof.push_back(d); }
                else if (pdof[nbase + local_of[node]] != d) bad = true;
                ln.push_back((uint16_t)local_of[node]);
            }
        for (int32_t c : cells) ecell.push_back(c);
        const size_t nn = pnode.size() - nbase;
        if (nn >= 0xFFFF || cells.size() >= 0xFFFF) { set_error("vector patch plan: patch too large (%zu nodes)", nn); return TB_ERR_UNSUPPORTED; }
        hdr[4 * q] = (uint32_t)(ecell.size() - cells.size());
        hdr[4 * q + 1] = (uint32_t)nbase;
        if (!halo) hdr[4 * q + 2] = (uint32_t)nn; // every touched node is accumulated and added
        hdr[4 * q + 2] |= (uint32_t)nn << 16;
        hdr[4 * q + 3] = (uint32_t)cells.size();
        plan->max_nodes = std::max(plan->max_nodes, (int)nn);
        plan->max_elems = std::max(plan->max_elems, (int)cells.size());
        for (size_t k = nbase; k < pnode.size(); ++k) local_of[pnode[k]] = -1;
    }
    if (bad) { set_error("vector patch plan: dofs and vertices of the field are not in one-to-one correspondence"); return TB_ERR_UNSUPPORTED; }
    // the kernel keeps 32 B per patch node in LDS and is built for three workgroups per CU: fragmented patches (quantile buckets on hollow or
    // strongly graded meshes, a large TB_VPATCH_TILE) that need more than a third of the 160 KiB take the general kernels instead of a launch failure
    if ((int64_t)plan->max_nodes * 32 > 160 * 1024 / 3) {
        set_error("vector patch plan: a patch touches %d nodes (%d B of LDS, more than a third of a CU's 160 KiB)", plan->max_nodes, plan->max_nodes * 32);
        return TB_ERR_UNSUPPORTED;
    }
    if ((int64_t)ecell.size() >= (int64_t)0x7fffffff || (int64_t)pnode.size() >= (int64_t)0x7fffffff) { set_error("vector patch plan: too many instances for 32-bit offsets"); return TB_ERR_UNSUPPORTED; }
    plan->total_elems = (int64_t)ecell.size();
    plan->total_nodes = (int64_t)pnode.size();
    std::vector<double> pcoord(pnode.size() * 3);
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < (int64_t)pnode.size(); ++k)
        for (int d = 0; d < 3; ++d) pcoord[3 * k + d] = m->h_xyz[3 * (int64_t)pnode[k] + d];
    if (getenv("TB_PLAN_VERBOSE"))
        fprintf(stderr, "[tbhip] vector patch plan (%s): %lld patches, %.3f instances per cell, %.3f patch nodes per dof, max %d instances / %d nodes per patch\n",
                halo ? "halo" : "own cells", (long long)np, (double)ecell.size() / (double)std::max<int64_t>(nc, 1), (double)pnode.size() / (double)std::max<int64_t>(m->ndofs, 1),
                plan->max_elems, plan->max_nodes);
    int rc = TB_OK;
    if (!rc) rc = upload(m->dev, hdr, &plan->d_hdr);
    if (!rc) rc = upload(m->dev, ln, &plan->d_elem_ln);
    if (!rc) rc = upload(m->dev, ecell, &plan->d_elem_cell);
    if (!rc) rc = upload(m->dev, pcoord, &plan->d_pcoord);
    if (!rc) rc = upload(m->dev, pdof, &plan->d_pdof);
    if (rc) { hipFree(plan->d_hdr); hipFree(plan->d_elem_ln); hipFree(plan->d_elem_cell); hipFree(plan->d_pcoord); hipFree(plan->d_pdof); return rc; }
    m->vpatches[halo] = std::move(plan);
    return TB_OK;
}

} // namespace tb
