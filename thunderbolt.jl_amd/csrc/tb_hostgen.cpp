// tb_hostgen.cpp — host-side generators of Ferrite-convention inputs (no GPU): structured hex grids
// (generate_grid, wrapped by src/mesh/generators.jl:942), close!(dh) dof numbering and the
// allocate_matrix sparsity pattern.  Used by bench.py / tests to build synthetic meshes at the
// BASELINE sizes; a Julia host passes Ferrite's own arrays instead.
//
// Algorithms are deliberately data-parallel (min-key + prefix sum, per-row merges) rather than the
// sequential first-visit / set-insertion form of the library they mimic.
#include <algorithm>
#include <parallel/algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <vector>

#include "tb_internal.h"

using namespace tb;

namespace {

constexpr int HEX_EDGES[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6}, {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
constexpr int HEX_FACES[6][4] = {{0, 3, 2, 1}, {0, 1, 5, 4}, {1, 2, 6, 5}, {2, 3, 7, 6}, {0, 4, 7, 3}, {4, 5, 6, 7}};

struct EntityRef {
    std::array<int32_t, 4> key; // sorted vertex ids (unused = -1)
    int64_t slot;               // cell*slots_per_cell + local entity slot  (visiting order)
    bool operator<(const EntityRef &o) const { return key != o.key ? key < o.key : slot < o.slot; }
};

} // namespace

extern "C" {

int tb_host_generate_grid_hex(int nx, int ny, int nz, const double *left, const double *right, double *xyz, int32_t *conn)
{
    TB_REQUIRE(nx > 0 && ny > 0 && nz > 0 && left && right && xyz && conn, "tb_host_generate_grid_hex: bad argument");
    const int64_t px = nx + 1, py = ny + 1, pz = nz + 1;
    TB_REQUIRE(px * py * pz < 0x7fffffff, "tb_host_generate_grid_hex: too many nodes for Int32");
    const int n[3] = {nx, ny, nz};
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < pz; ++k)
        for (int64_t j = 0; j < py; ++j)
            for (int64_t i = 0; i < px; ++i) {
                const int64_t id = i + px * (j + py * k);
                const int64_t ijk[3] = {i, j, k};
                for (int d = 0; d < 3; ++d) {
                    // lattice point on [left,right]: exact at both ends
                    const double v = left[d] + (right[d] - left[d]) * ((double)ijk[d] / n[d]);
                    xyz[3 * id + d] = ijk[d] == n[d] ? right[d] : v;
                }
            }
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < nz; ++k)
        for (int64_t j = 0; j < ny; ++j)
            for (int64_t i = 0; i < nx; ++i) {
                int32_t *e = conn + 8 * (i + (int64_t)nx * (j + (int64_t)ny * k));
                const int64_t b = i + px * (j + py * k);
                e[0] = (int32_t)b;             e[1] = (int32_t)(b + 1);
                e[2] = (int32_t)(b + 1 + px);  e[3] = (int32_t)(b + px);
                e[4] = (int32_t)(b + px * py); e[5] = (int32_t)(b + 1 + px * py);
                e[6] = (int32_t)(b + 1 + px + px * py); e[7] = (int32_t)(b + px + px * py);
            }
    return TB_OK;
}

// smooth interior perturbation (keeps the boundary planes): defeats affine-cell shortcuts
int tb_host_perturb_nodes(int nx, int ny, int nz, double amplitude_rel, double *xyz)
{
    TB_REQUIRE(nx > 0 && ny > 0 && nz > 0 && xyz, "tb_host_perturb_nodes: bad argument");
    const int64_t px = nx + 1, py = ny + 1, pz = nz + 1;
    const int64_t last = px * py * pz - 1;
    const double h[3] = {(xyz[3 * last] - xyz[0]) / nx, (xyz[3 * last + 1] - xyz[1]) / ny, (xyz[3 * last + 2] - xyz[2]) / nz};
    const double twopi = 6.283185307179586;
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < pz; ++k)
        for (int64_t j = 0; j < py; ++j)
            for (int64_t i = 0; i < px; ++i) {
                const int64_t id = i + px * (j + py * k);
                const double s = std::sin(twopi * i / nx) * std::sin(twopi * j / ny) * std::sin(twopi * k / nz);
                xyz[3 * id + 0] += amplitude_rel * h[0] * s;
                xyz[3 * id + 1] -= 0.5 * amplitude_rel * h[1] * s;
                xyz[3 * id + 2] += 0.75 * amplitude_rel * h[2] * s;
            }
    return TB_OK;
}

int tb_host_generate_grid_quad(int nx, int ny, const double *left, const double *right, double *xyz, int32_t *conn)
{
    if (!(nx > 0 && ny > 0 && left && right && xyz && conn)) { set_error("tb_host_generate_grid_quad: bad argument"); return TB_ERR_BAD_ARG; }
    const int px = nx + 1, py = ny + 1;
    // node lattice by corner interpolation like Ferrite's generate_grid (x fastest)
    for (int j = 0; j < py; ++j)
        for (int i = 0; i < px; ++i) {
            const double tx = (double)i / nx, ty = (double)j / ny;
            double *p = xyz + 3 * ((int64_t)j * px + i);
            p[0] = left[0] + (right[0] - left[0]) * tx;
            p[1] = left[1] + (right[1] - left[1]) * ty;
            p[2] = 0.0;
        }
    for (int j = 0; j < ny; ++j)
        for (int i = 0; i < nx; ++i) {
            int32_t *c = conn + 4 * ((int64_t)j * nx + i);
            const int32_t n0 = (int32_t)(j * px + i);
            c[0] = n0; c[1] = n0 + 1; c[2] = n0 + 1 + px; c[3] = n0 + px;
        }
    return TB_OK;
}

int64_t tb_host_close_dofs(int field_kind, int ncomp, int64_t n_cells, int64_t n_nodes, const int32_t *conn, int32_t *cell_dofs)
{
    if (!(conn && cell_dofs && n_cells >= 0 && n_nodes > 0 && ncomp >= 1) ||
        !(field_kind == TB_HEX8 || field_kind == TB_TET4 || field_kind == TB_HEX27 || field_kind == TB_QUAD4)) {
        set_error("tb_host_close_dofs: bad argument");
        return TB_ERR_BAD_ARG;
    }
    const int nv = kind_nverts(field_kind == TB_HEX27 ? TB_HEX8 : field_kind);
    const int slots = field_kind == TB_HEX27 ? 27 : nv; // entity slots per cell, in visiting order
    const int nb = slots;
    // first visit of every entity = smallest (cell, slot) key that references it
    std::vector<int64_t> first_of_slot((size_t)n_cells * slots); // per (cell,slot): key of the entity's first visit
    {
        std::vector<int64_t> first(n_nodes, INT64_MAX);
        for (int64_t c = 0; c < n_cells; ++c)
            for (int a = 0; a < nv; ++a) {
                int64_t &f = first[conn[c * nv + a]];
                f = std::min(f, c * slots + a);
            }
#pragma omp parallel for schedule(static)
        for (int64_t c = 0; c < n_cells; ++c)
            for (int a = 0; a < nv; ++a) first_of_slot[c * slots + a] = first[conn[c * nv + a]];
    }
    if (field_kind == TB_HEX27) {
        auto resolve = [&](std::vector<EntityRef> &refs) {
            std::sort(refs.begin(), refs.end());
            int64_t cur = -1;
            for (size_t i = 0; i < refs.size(); ++i) {
                if (i == 0 || refs[i].key != refs[i - 1].key) cur = refs[i].slot;
                first_of_slot[refs[i].slot] = cur;
            }
        };
        std::vector<EntityRef> refs((size_t)n_cells * 12);
#pragma omp parallel for schedule(static)
        for (int64_t c = 0; c < n_cells; ++c)
            for (int e = 0; e < 12; ++e) {
                int32_t a = conn[c * 8 + HEX_EDGES[e][0]], b = conn[c * 8 + HEX_EDGES[e][1]];
                if (a > b) std::swap(a, b);
                refs[c * 12 + e] = {{a, b, -1, -1}, c * 27 + 8 + e};
            }
        resolve(refs);
        refs.resize((size_t)n_cells * 6);
#pragma omp parallel for schedule(static)
        for (int64_t c = 0; c < n_cells; ++c)
            for (int f = 0; f < 6; ++f) {
                std::array<int32_t, 4> k = {conn[c * 8 + HEX_FACES[f][0]], conn[c * 8 + HEX_FACES[f][1]], conn[c * 8 + HEX_FACES[f][2]],
                                            conn[c * 8 + HEX_FACES[f][3]]};
                std::sort(k.begin(), k.end());
                refs[c * 6 + f] = {k, c * 27 + 20 + f};
            }
        resolve(refs);
        for (int64_t c = 0; c < n_cells; ++c) first_of_slot[c * 27 + 26] = c * 27 + 26;
    }
    // rank of each first-visit key = exclusive prefix sum over "is a first visit" flags
    std::vector<int32_t> rank((size_t)n_cells * slots + 1, 0);
    for (int64_t s = 0; s < n_cells * slots; ++s) rank[s + 1] = rank[s] + (first_of_slot[s] == s ? 1 : 0);
    const int64_t nent = rank[(size_t)n_cells * slots];
#pragma omp parallel for schedule(static)
    for (int64_t c = 0; c < n_cells; ++c)
        for (int a = 0; a < nb; ++a)
            for (int k = 0; k < ncomp; ++k)
                cell_dofs[(c * nb + a) * ncomp + k] = rank[first_of_slot[c * slots + a]] * ncomp + k;
    return nent * ncomp;
}

int64_t tb_host_build_pattern(int64_t n_cells, int ndpc, const int32_t *cell_dofs, int64_t ndofs, int64_t *rowptr, int32_t *colidx)
{
    if (!(cell_dofs && rowptr && n_cells >= 0 && ndofs > 0 && ndpc > 0)) {
        set_error("tb_host_build_pattern: bad argument");
        return TB_ERR_BAD_ARG;
    }
    // dof → cells
    std::vector<int64_t> ptr(ndofs + 1, 0);
    for (int64_t i = 0; i < n_cells * ndpc; ++i) ptr[cell_dofs[i] + 1]++;
    for (int64_t d = 0; d < ndofs; ++d) ptr[d + 1] += ptr[d];
    std::vector<int64_t> pos(ptr.begin(), ptr.end() - 1);
    std::vector<int32_t> cells((size_t)n_cells * ndpc);
    for (int64_t c = 0; c < n_cells; ++c)
        for (int l = 0; l < ndpc; ++l) cells[pos[cell_dofs[c * ndpc + l]]++] = (int32_t)c;
    auto row_columns = [&](int64_t r, std::vector<int32_t> &buf) {
        buf.clear();
        for (int64_t k = ptr[r]; k < ptr[r + 1]; ++k) {
            const int32_t *d = cell_dofs + (int64_t)cells[k] * ndpc;
            buf.insert(buf.end(), d, d + ndpc);
        }
        std::sort(buf.begin(), buf.end());
        buf.erase(std::unique(buf.begin(), buf.end()), buf.end());
    };
    if (!colidx) {
        rowptr[0] = 0;
#pragma omp parallel
        {
            std::vector<int32_t> buf;
#pragma omp for schedule(static)
            for (int64_t r = 0; r < ndofs; ++r) { row_columns(r, buf); rowptr[r + 1] = (int64_t)buf.size(); }
        }
        for (int64_t r = 0; r < ndofs; ++r) rowptr[r + 1] += rowptr[r];
        return rowptr[ndofs];
    }
#pragma omp parallel
    {
        std::vector<int32_t> buf;
#pragma omp for schedule(static)
        for (int64_t r = 0; r < ndofs; ++r) {
            row_columns(r, buf);
            std::copy(buf.begin(), buf.end(), colidx + rowptr[r]);
        }
    }
    return rowptr[ndofs];
}

/* Locality order for an arbitrarily numbered mesh (round 6).  The reference's CPU strategies are insensitive to how cells and dofs are numbered
 * (a plain loop over the cells, src/modeling/core/coordinate_systems.jl:145-171); the device kernels are not — plans share scatter / row signatures
 * and SpMV gathers stay inside the caches only where neighbours carry nearby numbers.  The order produced here is the sweep over the per-axis cell
 * LAYERS the patch planner cuts (build_patch_plan: cells ranked by centroid per axis, R = extent ÷ mean cell extent equal-count buckets — exactly the
 * (i, j, k) layers of a possibly distorted structured grid, density-adaptive elsewhere): cells lexicographic in (k, j, i) with ties broken by the
 * centroid, nodes the same over R + 1 node layers, and dofs in the order of their first visit when the cells are traversed in the new order with the
 * local dofs in cell_dofs order — what close!(dh) produces on a grid stored in that order, so a lattice under any numbering gets generate_grid's own.
 * (Round 6 also measured a TILE-major order — dofs numbered patch by patch, so that every patch writes one contiguous 32 KB piece of each value array
 * instead of one piece per grid line: slower at 216³, fused M + K 1.58 → 1.67 ms and CG iteration 0.70 → 0.82 ms, profiles/r06_v1/ab_tile_order.log —
 * and removed it.) */
int tb_host_locality_permutation(int geom_kind, int64_t n_nodes, const double *xyz, int64_t n_cells, const int32_t *conn, int ndofs_per_cell,
                                 const int32_t *cell_dofs, int64_t ndofs, int index_base, int32_t *cell_perm, int32_t *node_perm, int32_t *dof_perm)
{
    TB_REQUIRE(xyz && conn && n_nodes > 0 && n_cells > 0, "tb_host_locality_permutation: bad argument");
    TB_REQUIRE(geom_kind == TB_QUAD4 || geom_kind == TB_HEX8 || geom_kind == TB_TET4, "tb_host_locality_permutation: geom_kind must be TB_QUAD4, TB_HEX8 or TB_TET4");
    TB_REQUIRE(index_base == 0 || index_base == 1, "tb_host_locality_permutation: index_base must be 0 or 1");
    TB_REQUIRE(!dof_perm || (cell_dofs && ndofs_per_cell > 0 && ndofs > 0), "tb_host_locality_permutation: dof_perm needs cell_dofs, ndofs_per_cell and ndofs");
    TB_REQUIRE(n_cells < 0x7fffffff && n_nodes < 0x7fffffff && ndofs < 0x7fffffff, "tb_host_locality_permutation: too large for Int32");
    const int nv = geom_kind == TB_HEX8 ? 8 : 4;
    const int64_t nc = n_cells;
    // centroids, extents (per axis: layer spacing of a hexahedron from its face-centre differences, bounding box otherwise), domain box
    std::vector<double> cen((size_t)nc * 3);
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300}, hsum[3] = {0, 0, 0};
    static const int lo_hi[3][2][4] = {{{0, 3, 4, 7}, {1, 2, 5, 6}}, {{0, 1, 4, 5}, {3, 2, 7, 6}}, {{0, 1, 2, 3}, {4, 5, 6, 7}}};
#pragma omp parallel for schedule(static) reduction(min : lo[:3]) reduction(max : hi[:3]) reduction(+ : hsum[:3])
    for (int64_t c = 0; c < nc; ++c) {
        double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300}, sum[3] = {0, 0, 0};
        const int32_t *cn = conn + c * nv;
        for (int a = 0; a < nv; ++a)
            for (int d = 0; d < 3; ++d) {
                const double v = xyz[3 * (int64_t)(cn[a] - index_base) + d];
                sum[d] += v; mn[d] = std::min(mn[d], v); mx[d] = std::max(mx[d], v);
            }
        double ext[3] = {mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]};
        if (nv == 8) {
            ext[0] = ext[1] = ext[2] = 0.0;
            for (int k = 0; k < 3; ++k) {
                double e[3] = {0, 0, 0};
                for (int a = 0; a < 4; ++a)
                    for (int d = 0; d < 3; ++d) e[d] += 0.25 * (xyz[3 * (int64_t)(cn[lo_hi[k][1][a]] - index_base) + d] - xyz[3 * (int64_t)(cn[lo_hi[k][0][a]] - index_base) + d]);
                for (int d = 0; d < 3; ++d) ext[d] = std::max(ext[d], std::fabs(e[d]));
            }
        }
        for (int d = 0; d < 3; ++d) {
            cen[3 * c + d] = sum[d] / nv;
            lo[d] = std::min(lo[d], mn[d]); hi[d] = std::max(hi[d], mx[d]); hsum[d] += ext[d];
        }
    }
    int64_t Rv[3];
    for (int d = 0; d < 3; ++d) {
        const double hmean = hsum[d] / (double)nc;
        int64_t R = hmean > 0 ? (int64_t)std::llround((hi[d] - lo[d]) / hmean) : 1;
        Rv[d] = std::min<int64_t>(std::max<int64_t>(R, 1), 1 << 20);
    }
    // layer of every item per axis: equal-count buckets of the coordinate ranks; then one sort by (layer z, layer y, layer x, x-coordinate, id)
    auto sweep_order = [&](int64_t n, auto coord, const int64_t (&R)[3], int32_t *out) {
        std::vector<uint32_t> layer((size_t)n * 3);
        for (int d = 0; d < 3; ++d) {
            std::vector<std::pair<double, int32_t>> by(n);
#pragma omp parallel for schedule(static)
            for (int64_t i = 0; i < n; ++i) by[i] = {coord(i, d), (int32_t)i};
            __gnu_parallel::sort(by.begin(), by.end());
#pragma omp parallel for schedule(static)
            for (int64_t r = 0; r < n; ++r) layer[3 * (size_t)by[r].second + d] = (uint32_t)((r * R[d]) / n);
        }
        struct Key { uint32_t k, j, i; double x; int32_t id; };
        std::vector<Key> keys(n);
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; ++i) keys[i] = {layer[3 * i + 2], layer[3 * i + 1], layer[3 * i], coord(i, 0), (int32_t)i};
        __gnu_parallel::sort(keys.begin(), keys.end(), [](const Key &a, const Key &b) {
            if (a.k != b.k) return a.k < b.k;
            if (a.j != b.j) return a.j < b.j;
            if (a.i != b.i) return a.i < b.i;
            if (a.x != b.x) return a.x < b.x;
            return a.id < b.id;
        });
        for (int64_t i = 0; i < n; ++i) out[i] = keys[i].id;
    };
    std::vector<int32_t> corder((size_t)nc);
    sweep_order(nc, [&](int64_t c, int d) { return cen[3 * c + d]; }, Rv, corder.data());
    if (cell_perm)
        for (int64_t k = 0; k < nc; ++k) cell_perm[k] = corder[k] + index_base;
    if (node_perm) {
        const int64_t Rn[3] = {Rv[0] + 1, Rv[1] + 1, geom_kind == TB_QUAD4 ? 1 : Rv[2] + 1};
        std::vector<int32_t> norder((size_t)n_nodes);
        sweep_order(n_nodes, [&](int64_t i, int d) { return xyz[3 * i + d]; }, Rn, norder.data());
        for (int64_t k = 0; k < n_nodes; ++k) node_perm[norder[k]] = (int32_t)k + index_base; // node_perm[old] = new
    }
    if (dof_perm) {
        std::fill(dof_perm, dof_perm + ndofs, (int32_t)-1);
        int32_t next = 0;
        for (int64_t k = 0; k < nc; ++k) {
            const int32_t *cd = cell_dofs + (int64_t)corder[k] * ndofs_per_cell;
            for (int a = 0; a < ndofs_per_cell; ++a) {
                const int64_t d = (int64_t)cd[a] - index_base;
                TB_REQUIRE(d >= 0 && d < ndofs, "tb_host_locality_permutation: cell_dofs entry out of range");
                if (dof_perm[d] < 0) dof_perm[d] = next++;
            }
        }
        for (int64_t d = 0; d < ndofs; ++d) { // dofs no cell touches keep their relative order behind the visited ones
            if (dof_perm[d] < 0) dof_perm[d] = next++;
            dof_perm[d] += index_base;
        }
    }
    return TB_OK;
}

} // extern "C"
