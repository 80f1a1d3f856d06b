// tb_internal.h — private declarations shared by the host objects and the HIP kernel launchers.
// Nothing here is part of the ABI (include/tbhip.h is).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "tbhip.h"

namespace tb {

void set_error(const char *fmt, ...);
// Tuning and comparison switches (launch shapes, kernels that were measured slower than the ones that ship) exist in the profiling build only
// (`make ablation` → libtbhip_ablation.so, -DTB_ABLATION): the product library does not read them, so what they select is not product code.
// The switches the product library does read are listed in include/tbhip.h ("Environment"), each with a test.
#ifdef TB_ABLATION
inline const char *tune_env(const char *name) { return getenv(name); }
#else
inline const char *tune_env(const char *) { return nullptr; }
#endif
void set_last_kernel(const char *fmt, ...); // tb_last_kernel_name: the instance an assembly call launched
const char *last_kernel();

#define TB_HIP(call)                                                                          \
    do {                                                                                      \
        hipError_t e__ = (call);                                                              \
        if (e__ != hipSuccess) {                                                              \
            tb::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
            return TB_ERR_HIP;                                                                \
        }                                                                                     \
    } while (0)

#define TB_REQUIRE(cond, ...)             \
    do {                                  \
        if (!(cond)) {                    \
            tb::set_error(__VA_ARGS__);   \
            return TB_ERR_BAD_ARG;        \
        }                                 \
    } while (0)

// A call that waits for the device or reads back, made while a graph capture is open, would invalidate the capture — and HIP 7.0 then keeps refusing
// work on that stream, which may be the host's own (tb_device_set_stream).  Such calls refuse BEFORE they touch the stream: the capture stays valid
// (tb_graph_end still returns a graph of what was enqueued so far) and the stream stays usable.
#define TB_NO_CAPTURE(dev_)                                                                                                                        \
    do {                                                                                                                                           \
        if ((dev_)->capturing) {                                                                                                                   \
            tb::set_error("%s: this call waits for the device or copies to / from the host; it cannot run while a graph capture is open "         \
                          "(tb_graph_begin ... tb_graph_end) - make it before the capture (first use of a plan) or after it", __func__);          \
            return TB_ERR_BAD_ARG;                                                                                                                 \
        }                                                                                                                                          \
    } while (0)
#define TB_SYNC_STREAM(dev_)                              \
    do {                                                  \
        TB_NO_CAPTURE(dev_);                              \
        TB_HIP(hipStreamSynchronize((dev_)->stream));     \
    } while (0)

// TB_PLAN_VERBOSE=1: wall time of the host-side plan builders, stage by stage, on stderr (what `setup_s.first_step_incl_plan_build` of bench.py is made of)
struct PlanTimer {
    const char *name;
    double t0;
    bool on;
    static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
    explicit PlanTimer(const char *n) : name(n), t0(0.0), on(getenv("TB_PLAN_VERBOSE") != nullptr) { if (on) t0 = now(); }
    void lap(const char *what) { if (on) { const double t = now(); fprintf(stderr, "[tbhip] plan time: %-28s %-34s %8.3f s\n", name, what, t - t0); t0 = t; } }
    ~PlanTimer() { lap("(rest)"); }
};


// status word written by kernels (device) and read back after each launch group
struct Status {
    int neg_detj;        // != 0: some cell had detJ <= 0
    int pattern_missing; // != 0: a coupling was not found in the CSR pattern
    long long cell;      // one offending cell (0-based)
};

inline int kind_nverts(int kind) { return kind == TB_TET4 || kind == TB_QUAD4 ? 4 : 8; }
inline int kind_nbasis(int kind) { return kind == TB_TET4 || kind == TB_QUAD4 ? 4 : kind == TB_HEX27 ? 27 : 8; }
inline int kind_order(int kind) { return kind == TB_HEX27 ? 2 : 1; }

// ---- assembly plans (built lazily on the host, cached on the mesh / pattern) ----
struct ColorPlan {
    int ncolors = 0;
    std::vector<int64_t> offsets; // ncolors+1, into d_cells
    int32_t *d_cells = nullptr;   // cells grouped by colour
};

struct EAPlan {                // dof → contributing (cell, local) slots, cell-ordered
    int64_t *d_ptr = nullptr;  // ndofs+1
    int32_t *d_src = nullptr;  // n_cells*ndpc indices into the element-vector buffer
    double *d_ea = nullptr;    // n_cells*ndpc element vectors
    int32_t *d_ell = nullptr;  // the same slots as a fixed-width table (ell_w per dof, −1 padded), built on first use by ensure_ea_ell
    int32_t *d_ell_t = nullptr; // scalar Q2: the table with the local row of every slot replaced by its tensor index i₀ + 3 i₁ + 9 i₂ (ensure_ea_ell_q2t)
    int ell_w = 0;
    std::vector<int32_t> h_done; // per dof (scalar fields; built with d_ell): running maximum of the last contributing cell — the dofs before the first one whose
                                 // value reaches c are complete once cells [0, c) are integrated (chunked element assembly: gather behind the integration front)
};

// Patch plan: cells in Morton order are cut into patches; every dof (row) is owned by exactly one
// patch, which (re)computes all cells touching its rows, accumulates in LDS and stores once.
struct PatchPlan {
    int cells_per_patch = 0;
    int version = 0;   // bumped when the plan is rebuilt (LDS-fit retry); matrix extensions remember theirs
    int shrink = 0;    // tile reductions applied so far
    int64_t n_patches = 0;
    int max_elems = 0, max_rows = 0, threads = 256; // threads: workgroup size used by the patch kernels
    int64_t total_elems = 0, total_rows = 0;
    // per patch
    int64_t *d_elem_ptr = nullptr; // n_patches+1 → element instances
    int64_t *d_row_ptr = nullptr;  // n_patches+1 → owned rows
    // per element instance
    int32_t *d_elem_cell = nullptr; // global cell id
    uint16_t *d_elem_lrow = nullptr; // ndpc local-row slots per instance (0xFFFF = row not owned here)
    // per owned row
    int32_t *d_row_dof = nullptr;   // global dof id
    // matrix extension (needs a pattern)
    std::vector<int64_t> h_elem_ptr, h_row_ptr;
    std::vector<int32_t> h_elem_cell, h_row_dof;
    std::vector<uint16_t> h_elem_lrow;
};

struct RowDesc {      // one owned row of a patch
    int64_t nz0;      // first nz of the row in the global CSR arrays
    uint32_t off;     // offset of its accumulators in the patch's LDS block (in entries)
    uint32_t len;     // row length
};

struct PatchMatPlan {
    int version = 0;
    int max_lds_entries = 0;           // max over patches of Σ rowlen of owned rows
    RowDesc *d_row_desc = nullptr;     // per owned row
    uint16_t *d_elem_rowoff = nullptr; // per element instance × local dof: LDS offset of that row, 0xFFFF = not owned here
    uint8_t *d_colpos8 = nullptr;      // per element instance: ndpc*ndpc positions inside the row (rowlen <= 255)
    uint16_t *d_colpos16 = nullptr;    // same, 16-bit, when some row is longer
};

// Plan of the sum-factorised hexahedron patch kernel (tb_patch_fused.hip): per patch the list of the nodes its cell
// instances touch (owned rows first, in row-slot order), per instance 8 patch-local node indices (16 B: they select both
// the coordinates and the row accumulators) and the index of the cell's 64-byte signature "position of column dof(j) in row
// dof(i)" in a de-duplicated table (structured regions share a handful of signatures; an unstructured mesh keeps one per cell).
struct PatchFusedPlan {
    int version = 0;
    int max_lds_entries = 0, max_nodes = 0;
    int64_t nsig = 0;
    int64_t *d_node_ptr = nullptr; // n_patches+1
    int32_t *d_pnode = nullptr;    // global node ids
    uint16_t *d_elem_ln = nullptr; // 8 per instance
    uint32_t *d_elem_sig = nullptr;
    uint8_t *d_sigtab = nullptr;   // nsig × 64
    RowDesc *d_row_desc = nullptr;
    // staged kernels: packed patch headers (limits of the packed word: < 1024 rows, < 2048 nodes, < 2048 instances per patch) and pre-gathered vertex coordinates
    uint32_t *d_hdr = nullptr;     // 4 per patch: e0, r0, n0, nrows | nnodes << 10 | ne << 21 (NULL when some count does not fit)
    double *d_pcoord = nullptr;    // 3 per patch node
    // one-trip kernel (k_patch_hex8_record): every input of a patch in ONE record at blockIdx × stride, so no load waits for a header —
    //   [0,16) {nrows | nnodes << 10 | ne << 21, first instance, 0, 0} · rec_ne × 16 B local node indices · rec_ne × 4 B signature ids ·
    //   rec_rm × 16 B row descriptors · rec_nm × 24 B vertex coordinates; zero padding behind the patch's own counts.  Built on first use.
    uint8_t *d_rec = nullptr;
    int rec_stride = 0, rec_rm = 0, rec_nm = 0, rec_ne = 0; // rec_ne: instance slots per record (≥ 256, multiple of 64); rec_stride = −1: the patches do not fit the record form (a patch with more than 256 instances)
};

// Patch plan of the linear-form (vector) kernels on trilinear hexahedra (tb_assembly.hip: k_vector_hex8_patch).  Row accumulators of a vector
// are one double per node, so patches are large (8×8×8 tiles).  halo = true: rows owned by first touch, the patch re-integrates the halo cells
// and stores each owned dof once (deterministic, PATCH strategy); halo = false: own cells only, every touched node is added to the zeroed
// vector with one global atomic per patch node (ATOMIC strategy: 1.4 atomics per cell instead of 8).
struct VecPatchPlan {
    bool halo = false;
    int64_t n_patches = 0, total_elems = 0, total_nodes = 0;
    int max_nodes = 0, max_elems = 0;
    uint32_t *d_hdr = nullptr;      // 4 per patch: e0, n0, nrows | nnodes << 16, ne
    uint16_t *d_elem_ln = nullptr;  // 8 patch-local node indices per instance
    int32_t *d_elem_cell = nullptr; // global cell of every instance (tabulated sources, error reports)
    double *d_pcoord = nullptr;     // 3 per patch node
    int32_t *d_pdof = nullptr;      // global dof of every patch node
};

} // namespace tb

struct tb_pattern;
struct tb_device {
    int id = 0;
    hipStream_t stream = nullptr;
    bool own_stream = true;
    int n_cu = 0;
    size_t hbm_bytes = 0;
    std::string name;
    tb::Status *d_status = nullptr;
    tb::Status *h_status = nullptr; // pinned
    bool defer_status = false;      // tb_device_defer_status: assembly calls return without reading the status block; tb_device_poll_status reads it
    void *d_scratch = nullptr;      // Float64 arena behind the *_f32 entry points (tb_f32.hip), grown on demand
    size_t scratch_bytes = 0;
    double *d_slots = nullptr;      // reduction slots (tb_algebra.hip: block_sum_slots): RED_GROUPS groups of 64 partial sums, 128 B apart, zero between uses
    double *d_tslot = nullptr;      // {t, cos 2πt}: where time-dependent kernels read the time while a graph capture is open (tb_graph.hip)
    bool capturing = false, defer_before_capture = false, tslot_used = false; // tslot_used: a captured launch was handed the slot
    hipStream_t aux_stream = nullptr; // second queue of the chunked mechanics linearisation (gather of chunk k beside the integration of chunk k + 1)
    hipEvent_t aux_ev[2] = {nullptr, nullptr};
};

struct tb_mesh {
    tb_device *dev = nullptr;
    int geom_kind = 0, field_kind = 0, ncomp = 1;
    int nverts = 0, nb = 0, ndpc = 0;
    int64_t n_nodes = 0, n_cells = 0, ndofs = 0;
    double *d_xyz = nullptr;
    int32_t *d_conn = nullptr;
    int32_t *d_cell_dofs = nullptr;
    int32_t *d_node_dof0 = nullptr; // vector fields: first dof of every field node
    uint8_t *d_rank27 = nullptr;    // 27-node vector fields: rank of every node of a cell by global dof (32 bytes per cell), see tb_mechanics.hip
    int rank27_ok = 0;              // 0 not examined, 1 built, −1 not applicable
    double *d_cell_xyz = nullptr;   // vertex coordinates per cell (nverts × 3, cell-major), built on first use by the block-per-cell kernels
    int64_t n_nodes_field = 0;
    std::vector<int32_t> h_node_dof0;
    std::vector<double> h_xyz;
    std::vector<int32_t> h_conn, h_cell_dofs;
    std::unique_ptr<tb::ColorPlan> colors;
    std::unique_ptr<tb::EAPlan> ea;
    std::unique_ptr<tb::PatchPlan> patches;
    int patch_rcb = 0;            // > 0: the matrix patch plan bisects the cells into leaves of this many cells instead of tiling per-axis buckets (tb_plans.cpp)
    bool patch_rcb_tried = false; // the bisection was compared with the tiles once
    int patch_tile_shrink = 0;    // tile reductions the LDS-fit loop applied to the tile plan (restored when the bisection loses)
    std::unique_ptr<tb::VecPatchPlan> vpatches[2]; // [halo]
};

struct tb_pattern {
    tb_mesh *mesh = nullptr;
    int64_t n_rows = 0, nnz = 0;
    int64_t *d_rowptr = nullptr;
    int32_t *d_colidx = nullptr;
    std::vector<int64_t> h_rowptr;
    std::vector<int32_t> h_colidx;
    bool map64 = false;
    void *d_emap = nullptr; // [ndpc*ndpc][n_cells] nz index of (cell,i,j): int32 (nnz < 2^31) or int64
    double *d_cg_ws = nullptr;      // CG workspace (r, p, Ap, D⁻¹, 2 scalars)
    const double *cg_dinv_of = nullptr; // the nz array whose diagonal the D⁻¹ slot of d_cg_ws holds (TB_JACOBI_REUSE re-extracts for any other)
    double last_tol = 0.0;          // atol + rtol·‖r₀‖ of the latest Krylov solve on this pattern (tb_solver_last_tolerance)
    int32_t *d_bcol = nullptr;      // block SpMV of 3-dof-per-node patterns: one column (node) index per 3×3 block
    int b3 = 0, b3_lanes = 16;      // b3: 0 = not examined, 1 = CSR of 3×3 blocks, −1 = not
    int64_t *d_diagpos = nullptr;   // nz index of each row's diagonal entry (−1 if absent), built at the first Jacobi-preconditioned solve
    int32_t *d_blkrow = nullptr;    // stream SpMV: first row of each workgroup's run of rows (n_blk + 1 entries)
    uint32_t *d_blkrec = nullptr;   // the same runs as 16-byte records {first row, rows | entries << 16, first nz low, high} (k_spmv_stream_rec)
    int64_t n_blk = 0;              // 0 = not planned yet, −1 = a row exceeds the capacity (lanes-per-row kernel)
    // index-compressed stream SpMV: rows that hold the same column offsets (colidx[k] − row) share one signature; a row then needs 4 B (the
    // position of its signature in the table) instead of 4 B per non-zero
    uint32_t *d_rowsig = nullptr;   // per row: first entry of its signature in d_sigoff
    int32_t *d_sigoff = nullptr;    // signatures back to back: column offsets relative to the row
    int64_t n_sig = 0;              // 0 = not planned yet, −1 = the pattern does not compress (CSR kernel), > 0 = signatures in the table
    int64_t sig_entries = 0;
    // sliced mirror of one value array (tb_spmv_mirror): per slice of 64 consecutive rows the values as [entry k][lane], zero-padded to the longest
    // row of the slice — every value load of the product a coalesced 512-byte run, no LDS staging.  Column offsets come from the signature table.
    static constexpr int MIRRORS = 2; // arrays of one pattern that can be bound at a time (the system matrix of a solve and one more: K of the heat step's right-hand side)
    double *d_mir[MIRRORS] = {nullptr, nullptr};        // mirrored values (Σ 64 · width doubles each)
    void *d_mir_base = nullptr;     // n_slices + 1 slice records {first value, first column offset, shared signature or none, width}
    int32_t *d_mir_off = nullptr;   // column offsets of the slices of mixed signatures, entry-major like the values
    std::vector<uint32_t> h_rowsig; // host copy of d_rowsig (slice table)
    int64_t n_slices = 0;           // 0 = not planned, −1 = the pattern has no mirror (no signature plan, rows longer than 255 entries)
    int64_t mir_entries = 0;
    const double *mir_nz[MIRRORS] = {nullptr, nullptr}; // the value array each mirror currently reflects (products with this pointer use it); NULL = unbound
    uint64_t mir_stamp[MIRRORS] = {0, 0}, mir_clock = 0; // bind / refresh order: a new binding evicts the slot bound or refreshed longest ago (true LRU, what api.py keeps alive)
    uint32_t *d_wrunrec = nullptr;  // wave-private SpMV (TB_SPMV_KERNEL=wave): runs of ≤ 21 rows as 16-byte records
    int64_t n_wrun = 0;
    uint16_t *d_q2pos = nullptr;    // scalar Q2 forms: position of col dof(j) in row dof(i), per cell and pair
    uint16_t *d_q2pos_t = nullptr;  // the same in the order of the stored element matrices (element strategy: tensor order, k_matrix_q2_sf)
    double *d_pcg_ws = nullptr;     // general-preconditioner PCG workspace (r, z, p, Ap, D̃, scalars)
    double *d_cheb_ws = nullptr;    // Chebyshev-preconditioned CG workspace (r, z, p, Ap, D⁻¹, d, w, scalars)
    double *d_gmres_ws = nullptr;   // GMRES workspace: (restart+1) basis vectors + 3 vectors + scalars
    int gmres_m = 0;
    double *d_kebuf = nullptr;      // element-matrix buffer of the ElementAssemblyStrategy (vector fields)
    size_t kebuf_bytes = 0;
    double *d_qpbuf = nullptr;      // quadrature-point records of the split mechanics linearisation (k_mech_points → k_mech_contract), one launch's worth
    size_t qpbuf_bytes = 0;
    void *d_gnodes = nullptr;       // per field node: {first nz, row length, cell count, ≤ 8 element-matrix run offsets} of the staged gather
    int gnodes_state = 0;           // 0 not built, 1 built, −1 some node sits in more than 8 cells (direct gather is used)
    std::vector<int32_t> h_gn_last; // running maximum of the records' last contributing cell: the record prefix completed by a chunk of cells
    int64_t max_row_len = 0;
    uint16_t *d_blockpos = nullptr; // vector fields: per cell and node pair, position of the 3×3 block inside its row
    std::unique_ptr<tb::PatchMatPlan> patch_mat;
    std::unique_ptr<tb::PatchFusedPlan> patch_fused;
};

struct tb_form {
    tb_mesh *mesh = nullptr;
    int kind = 0, qorder = 0;
    tb_coef coef{};
    double Dconst[9] = {0}; // folded constant tensor (incl. κ/(Cₘχ))
    bool field = false, symmetric = true;
    double *d_field = nullptr;
    double *d_dtab = nullptr;  // diffusion tensor at the quadrature points (built from d_field at the first assembly)
    double *d_gtab = nullptr;  // trilinear hexahedra, field tensors: G_q = −¼/detJ·A·D_q·Aᵀ (6) and detJ_q per (cell, point) — mesh and field are fixed, so is G
    double *d_table = nullptr;
    int64_t table_len = 0;
    tb_material mat{};
    // active stress (TB_FORM_HYPERELASTIC): Ta = act_tension · (nodal field per cell, or 1)
    double act_tension = 0.0;
    double *d_act_field = nullptr;
    // subdomain (tb_form_set_cellset): the cells this form integrates over, and whether it adds to its outputs (tb_form_set_accumulate)
    bool has_cellset = false;
    int accumulate = 0;
    int64_t n_set = 0;
    int32_t *d_cellset = nullptr;
    std::unique_ptr<tb::ColorPlan> set_colors;
    std::vector<int32_t> h_cellset;
    // condensed internal variable (tb_hyperelastic_set_condensation / _set_internal_state)
    int cond_model = 0, cond_max_iters = 10;
    double cond_params[17] = {0}, cond_tmax = 0.0, cond_tol = 1e-4, cond_dt = 0.0;
    double *d_Q = nullptr;
    const double *d_Qknown = nullptr;
    const double *d_u_prev = nullptr; // rate-coupled local problem: previous displacement (NULL: rate-free)
    double *d_qp_buf = nullptr, *d_qp_act = nullptr; // λ | Ca | (a, b) | status per quadrature point
    int64_t cond_n_failed = 0;
    // PrestressedMechanicalModel: F₀⁻¹ (tb_hyperelastic_set_prestress)
    int prestressed = 0;
    double prestress_G[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    // Hill frameworks (tb_hyperelastic_set_hill)
    int hill = 0, act_energy = 0, act_penalty = 0, adg = 0, sarc = 0;
    double act_p[12] = {0}, hill_kappa = 0.0, sarc_p[2] = {0, 0};
    // weak boundary conditions (TB_FORM_FACET)
    int bc_kind = 0, facet_q = 0;
    double bc_param = 0.0;
    int32_t *d_facets = nullptr; // (cell, local facet) pairs, 0-based
    int64_t n_facets = 0;
};

namespace tb {

// ---- plan builders (tb_plans.cpp) ----
void dof_slots(const tb_mesh *m, std::vector<int64_t> &ptr, std::vector<int32_t> &src); // dof → (cell·ndpc + local) slots, cell-ordered
int build_color_plan(tb_mesh *m);
int build_color_plan_subset(tb_mesh *m, const std::vector<int32_t> &cells, std::unique_ptr<ColorPlan> &out);
int build_ea_plan(tb_mesh *m);
int ensure_ea_ell(tb_mesh *m);
int ensure_ea_ell_q2t(tb_mesh *m);
int build_patch_plan(tb_mesh *m, int cells_per_patch);
int build_patch_mat_plan(tb_pattern *p);
int ensure_patch_records(tb_pattern *p); // one-trip records of the fused plan (after ensure_patch_fused); TB_ERR_UNSUPPORTED when the patches do not fit the form
int ensure_patch_plans(tb_mesh *m, tb_pattern *p); // builds / refits both so the LDS block allows two workgroups per CU
void free_patch_plan(tb_mesh *m);
void free_patch_mat_plan(tb_pattern *p);
int ensure_vec_patch_plan(tb_mesh *m, bool halo);
void free_vec_patch_plans(tb_mesh *m);
int ensure_patch_fused(tb_mesh *m, tb_pattern *p, int nregions); // plan of the sum-factorised hexahedron kernel; nregions accumulator blocks must fit 80 KiB of LDS
void free_patch_fused_plan(tb_pattern *p);
int launch_assemble_hex8_patch(tb_form *fK, tb_form *fM, tb_pattern *p, double t, double *d_nzK, double *d_nzM); // either form may be NULL
bool hex8_patch_applicable(const tb_form *f, const tb_pattern *p);
int launch_assemble_tet4_patch(tb_form *fK, tb_form *fM, tb_pattern *p, double t, double *d_nzK, double *d_nzM); // linear tetrahedra; either form may be NULL
bool tet4_patch_applicable(const tb_form *f, const tb_pattern *p);

// ---- kernel launchers (tb_assembly.hip / tb_reaction.hip / tb_algebra.hip) ----
int ensure_emap(tb_pattern *p);
int launch_assemble_matrix(tb_form *f, tb_pattern *p, int strategy, double t, double *d_nz);
int launch_assemble_vector(tb_form *f, int strategy, double t, double *d_b);
int launch_hyperelastic(tb_form *f, tb_pattern *p, int strategy, const double *d_u, double *d_nz, double *d_r);
int host_material_eval_form(tb_form *form, const double *F9, double *psi, double *P, double *A);
int host_material_eval(const tb_material *mat, const double *F9, double *psi, double *P, double *A);
int launch_reaction(tb_device *dev, int model, const double *params, int n_params, double *d_u, double *d_du,
                    int64_t n_points, int layout, double t, double dt, int substeps, double thr, double *rmax /*nullable, host*/,
                    const float *d_x = nullptr, int sdim = 0);
int launch_reaction_f32(tb_device *dev, int model, const double *params, int n_params, float *d_u, float *d_du, int64_t n_points, int layout, double t, double dt,
                        int substeps, double thr, const float *d_x, int sdim);
int launch_reaction_rl(tb_device *dev, int model, const double *params, int n_params, double *d_u, int64_t n, int layout, double t, double dt);
int launch_heat_matrix(tb_device *dev, int64_t nnz, const double *M, const double *K, double dt, double *A);
int launch_spmv(tb_pattern *p, const double *nz, const double *x, double alpha, double beta, double *y);
int launch_sarcomere(tb_device *dev, const double *params, double *d_state, int64_t n, const double *d_stretch, const double *d_velocity,
                     const double *d_calcium, double stretch, double velocity, double calcium, double dt, int substeps, int rate_independent,
                     double *d_tension, double *d_stiffness);
int launch_sarcomere_implicit(tb_device *dev, const double *params, double *d_Q, const double *d_Qknown, int64_t n, const double *d_stretch,
                              const double *d_velocity, const double *d_calcium, double stretch, double velocity, double calcium, double dt, double tol,
                              int max_iters, double tmax, double *d_dQdl, double *d_dQdv, double *d_act, int act_stride, int32_t *d_status, int64_t *n_failed,
                              int need_sens, const int32_t *d_cells = nullptr, int nq = 1, int64_t n_cells_listed = 0);
int host_sarcomere_local_solve(const double *params, double *Q, const double *Qknown, double stretch, double velocity, double calcium, double dt, double tol,
                               int max_iters, double *dQdl, double *dQdv, int *iters, double *resnorm);
void host_sarcomere_derivatives(const double *params, const double *u, double stretch, double velocity, double calcium, int analytic, double *D, double *gl,
                                double *gv, double *rhs);
void host_sarcomere_eval(const double *params, const double *u, double stretch, double velocity, double calcium, double *du, double *tension,
                         double *stiffness);
int launch_l1gs_setup(tb_pattern *pat, const double *A, int ps, double *d_dtilde);
int launch_l1gs_apply(tb_pattern *pat, const double *A, const double *d_dtilde, int ps, int symmetric, const double *r, double *z);
int launch_pcg_chebyshev(tb_pattern *pat, const double *A, const double *b, double *x, double rtol, double atol, int maxiter, int degree, int *iters, double *resnorm);
int launch_pcg_l1gs(tb_pattern *pat, const double *A, const double *b, double *x, double rtol, double atol, int maxiter, int ps, int *iters, double *resnorm);
int launch_gmres(tb_pattern *pat, const double *A, const double *b, double *x, double rtol, double atol, int maxiter, int restart, int jacobi,
                 int *iters, double *resnorm);
int launch_cg(tb_pattern *pat, const double *A, const double *b, double *x, double rtol, double atol, int maxiter, int jacobi, int *iters,
              double *resnorm, bool b_is_residual = false);
int launch_axpy(tb_device *dev, int64_t n, double a, const double *x, double *y);
int launch_absmax(tb_device *dev, int64_t n, const double *x, int64_t stride, double *result);
int launch_facets(tb_form *f, tb_pattern *p, const double *d_u, double *d_nz, double *d_r);
int ensure_blockpos(tb_pattern *p);
int launch_apply_zero(tb_pattern *pat, double *nz, double *f, const uint8_t *flags, double diag);
int launch_meandiag(tb_pattern *pat, const double *nz, double *result);
int launch_dot(tb_device *dev, int64_t n, const double *a, const double *b, double *result);
int launch_cgd_dot(tb_device *dev, int64_t n, const double *w, const double *a, const double *b, double *d_out);
int launch_cgd_update(tb_device *dev, int64_t n, const double *w, const double *dinv, const double *p, const double *Ap, double *x, double *r,
                      const double *d_rz, const double *d_pAp, double *d_out3);
int launch_cgd_direction(tb_device *dev, int64_t n, const double *dinv, const double *r, double *p, const double *d_rz, const double *d_rz_new);
int launch_max(tb_device *dev, int64_t n, const double *x, int64_t stride, double *result);
int ensure_cell_xyz(tb_mesh *m); // tb_assembly.hip: builds tb_mesh::d_cell_xyz on first use
int launch_gather_indexed(tb_device *dev, int64_t n, const double *vec, const int32_t *idx, double *out);
int launch_scatter_add_indexed(tb_device *dev, int64_t n, const double *in, const int32_t *idx, double *vec);
int read_status_public(tb_device *dev); // status block → host, synchronises the stream (what check_status does when the status is not deferred)
int launch_cgd_rotate(tb_device *dev, double *d_S);
int launch_cgd_iteration(tb_pattern *pat, const double *A, const double *dinv, double *x, double *r, double *p, double *Ap, double *d_S);
int spmv_plans(tb_pattern *p); // builds the plans tb_spmv_csr would build on its first product
int launch_scatter_indexed(tb_device *dev, int64_t n, const double *in, const int32_t *idx, double *vec);
int launch_spmv_rows(tb_pattern *p, const double *nz, const double *x, int64_t n, const int32_t *rows, double *out);
int launch_mirror_bind(tb_pattern *p, const double *nz); // tb_spmv_mirror
int ensure_aux_stream(tb_device *dev); // second queue + two events of the chunked element assemblies (tb_mechanics.hip, scalar Q2 forms)
int launch_extract_diagonal(tb_pattern *p, const double *nz, double *diag);
int launch_spmv_dot(tb_pattern *pat, const double *A, const double *x, double *y, double *d_dot);
double decode_ordered_key(unsigned long long k);

int check_status(tb_device *dev);
int reset_status(tb_device *dev);

template <class T>
int upload(tb_device *dev, const std::vector<T> &h, T **d)
{
    *d = nullptr;
    if (h.empty()) return TB_OK;
    TB_NO_CAPTURE(dev); // a blocking upload of a host vector: plan building belongs in front of a capture
    TB_HIP(hipMalloc((void **)d, h.size() * sizeof(T)));
    TB_HIP(hipMemcpyAsync(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, dev->stream));
    TB_SYNC_STREAM(dev);
    return TB_OK;
}

} // namespace tb
