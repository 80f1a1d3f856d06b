/*
 * tbhip.h — C ABI of the MI355X (gfx950) hot-path backend for Thunderbolt.jl.
 *
 * This is the drop-in boundary: exactly what a Julia `ccall` binding (julia/ThunderboltHIPBackend.jl,
 * see INTEGRATION.md) needs to provide `MI355XDevice <: AbstractGPUDevice` behind Thunderbolt's
 * operator API.  Plain pointers and sizes only; no C++ types, no exceptions, every function returns
 * an `int` status (0 = TB_OK, <0 = error) and `tb_last_error_string()` describes the last failure
 * on the calling thread.
 *
 * Each entry cites the reference interface it replaces (paths relative to the Thunderbolt.jl
 * v0.0.4 tree).  The cell loop / scatter the kernels replace lives in FerriteOperators.jl (third
 * party); its visible call sites are cited instead.
 *
 * Pointer naming: `d_*` = device pointer (owned by the caller, e.g. a Julia GC-managed wrapper with
 * a finaliser calling tb_free), everything else = host pointer.  All indices handed in may be 0- or
 * 1-based (`index_base`); the library stores 0-based Int32 on device.
 * Output arrays of the assembly calls are OVERWRITTEN (the reference zero-fills before assembling:
 * src/solver/nonlinear/newton_raphson.jl:231, src/disambiguation.jl:26-35).
 *
 * Threading: calls on one tb_device are serialised on that device's HIP stream and are synchronous
 * on return only where stated; different devices may be driven from different threads.
 *
 * Environment: the library reads these variables and no others (each has a test in tests/):
 *   TB_PATCH_KERNEL = record | staged | general   patch kernel of first-order matrices (default: record where it applies; the others are what
 *                                                 field coefficients and oversize patches run — the switch lets a test put them on any mesh)
 *   TB_PATCH_CUT = full, TB_PATCH_TILE = "x,y,z"  tile cut / tile shape of the patch plan
 *   TB_PATCH_ISO = 0                              constant positive definite tensors through the DIAG / general instances instead of the ISO one
 *   TB_SPMV_KERNEL = rows                         CSR rows kernel also where the pattern compresses (what patterns without shared signatures run)
 *   TB_MECH_CHUNKS = n                            launches of the chunked Q2 linearisation (0 / 1: one)
 *   TB_PLAN_VERBOSE = 1                           plan statistics on stderr;   TB_RCCL_LIBRARY = path   the RCCL to open
 * Tuning and comparison switches of earlier rounds exist in the profiling build only (make -C thunderbolt.jl_amd/csrc ablation).
 */
#ifndef TBHIP_H
#define TBHIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tb_device tb_device;
typedef struct tb_mesh tb_mesh;
typedef struct tb_pattern tb_pattern;
typedef struct tb_form tb_form;

enum {
    TB_OK = 0,
    TB_ERR_BAD_ARG = -1,
    TB_ERR_HIP = -2,
    TB_ERR_NEG_DETJ = -3,    /* detJ <= 0 in some cell (src/ferrite-addons/PR883.jl:376 "TODO: return error code") */
    TB_ERR_PATTERN = -4,     /* a cell coupling is missing from the CSR pattern */
    TB_ERR_UNSUPPORTED = -5,
    TB_ERR_NOMEM = -6
};

/* cell kinds (Ferrite reference shapes; local vertex order as src/mesh/generators.jl:62-79) */
enum { TB_QUAD4 = 2 /* bilinear quadrilateral in the plane z = 0 (2-D problems; xyz still n×3) */, TB_HEX8 = 3, TB_TET4 = 4,
       TB_HEX27 = 5 /* Q2 field on trilinear HEX8 geometry */ };

/* assembly strategies — device analogues of the FerriteOperators strategies Thunderbolt re-exports
 * (src/Thunderbolt.jl:22-32; selected via FiniteElementDiscretization(; assembly_strategy), src/discretization/fem.jl:38-46) */
enum {
    TB_STRATEGY_ATOMIC = 0,    /* one thread per cell, FP64 hardware atomics                                   */
    TB_STRATEGY_PER_COLOR = 1, /* PerColorAssemblyStrategy: colours in sequence, plain read-modify-write       */
    TB_STRATEGY_ELEMENT = 2,   /* ElementAssemblyStrategy: element contributions summed per dof / row in cell   */
                               /* order, no atomics, bit-reproducible (forced for the source term by            */
                               /* src/solver/time/euler.jl:148-153).  Vectors and quadratic-field matrices store */
                               /* the element vectors / matrices and gather them; matrices of first-order fields */
                               /* run the per-colour kernels (one contribution per nz and colour, colours in     */
                               /* sequence: an ordered sum).  Two assemblies of the same form give identical     */
                               /* bits (tests: test_element_strategy_is_bit_reproducible).  The PATCH kernels    */
                               /* do NOT promise that: their LDS accumulators add in wave-scheduling order       */
    TB_STRATEGY_PATCH = 3      /* native: Morton patches of cells, rows accumulated in LDS, each nz / dof      */
                               /* written exactly once with coalesced stores (default, fastest)                */
};

/* bilinear / linear forms */
enum {
    TB_FORM_MASS = 0,      /* Mₑ[i,j] += ρ NᵢNⱼ dΩ          src/modeling/core/mass.jl:28-43       */
    TB_FORM_DIFFUSION = 1, /* Kₑ[i,j] -= ∇Nⱼ·D·∇Nᵢ dΩ       src/modeling/core/diffusion.jl:28-50  */
    TB_FORM_SOURCE = 2,    /* bₑ[j]  += f(x_q,t) Nⱼ dΩ      src/modeling/core/analytical_coefficient.jl:80-101 */
    TB_FORM_FACET = 4,       /* weak boundary conditions on hexahedron facets   src/modeling/core/weak_boundary_conditions.jl */
    TB_FORM_HYPERELASTIC = 3 /* rₑ[i] += ∇δuᵢ⊡P dΩ, Kₑ[i,j] += (∇δuᵢ⊡𝔸)⊡∇δuⱼ dΩ   src/modeling/solid/elements.jl:177-313 */
};

/* coefficients (src/modeling/core/coefficients.jl, src/modeling/microstructure.jl).  Julia closures cannot
 * cross a C ABI, so analytical coefficients are enumerated closed forms or host-tabulated values. */
enum {
    TB_COEF_CONST_SCALAR = 0,    /* p[0]: ρ, or isotropic D = p[0]·I         ConstantCoefficient, coefficients.jl:101-120 */
    TB_COEF_CONST_TENSOR = 1,    /* p[0..9) row-major 3×3                    ConstantCoefficient(Tensor)                  */
    TB_COEF_FIELD_SCALAR = 2,    /* field[basis + nb·cell]                   FieldCoefficient, coefficients.jl:85-99      */
    TB_COEF_SPECTRAL_CONST = 3,  /* p = f[3],s[3],n[3],λ[3]                  SpectralTensorCoefficient of constants, microstructure.jl:136-138 */
    TB_COEF_SPECTRAL_FIELD = 4,  /* field = per cell, per basis: f[3],s[3],n[3]; p = λ[3]; interpolated, normalised and
                                    Gram–Schmidt-orthogonalised per quadrature point  microstructure.jl:176-187, utils.jl:131-139 */
    TB_COEF_TRANSVERSE_CONST = 5 /* p = f[3], λ[2]:  λ₁ f⊗f + λ₂ (I − f⊗f)   microstructure.jl:89-92 */
};
enum {
    TB_SRC_CONST = 0,        /* p[0] */
    TB_SRC_NORM_PLUS_T = 1,  /* ‖x‖ + t               benchmarks/benchmarks-linear-form.jl:16-20 */
    TB_SRC_COS_EXP = 2,      /* cos(2πt)·exp(−‖x‖²)   test/gpu/test_operators.jl:13-18, benchmarks/benchmarks-cuda-linear-form.jl:15-18 */
    TB_SRC_TABULATED = 3     /* table[q + nq·cell], host-evaluated closure (tb_form_set_table) */
};

typedef struct tb_coef {
    int32_t kind;        /* TB_COEF_* (mass, diffusion) or TB_SRC_* (source) */
    int32_t wrap;        /* != 0: ConductivityToDiffusivityCoefficient, D = κ/(Cₘ·χ)  coefficients.jl:152-162, fem.jl:413-419 */
    double Cm, chi;
    double p[16];
    const double *field; /* HOST pointer, copied at tb_form_create; may be NULL */
    int64_t field_len;   /* number of doubles behind `field` */
} tb_coef;

/* constitutive models for the quasi-static path: PK1Model(material, microstructure) (src/modeling/solid/materials.jl:442-453) */
/* tb_material.kind: the passive energies of src/modeling/solid/energies.jl (parameters p[0…] in the struct field order of the reference) */
enum { TB_MATERIAL_HOLZAPFEL_OGDEN_2009 = 0, /* a, b, aᶠ, bᶠ, aˢ, bˢ, aᶠˢ, bᶠˢ                                  :136-168 */
       TB_MATERIAL_NULL = 1,                 /* Ψ = 0                                                            :6-7     */
       TB_MATERIAL_BIO_NEOHOOKEAN = 2,       /* α                                                                :461-473 */
       TB_MATERIAL_TI_NEOHOOKEAN = 3,        /* a₁, a₂, α₁, α₂ (TransverseIsotopicNeoHookeanModel)               :93-128  */
       TB_MATERIAL_LIN_YIN_PASSIVE = 4,      /* C₁…C₄                                                            :178-198 */
       TB_MATERIAL_LIN_YIN_ACTIVE = 5,       /* C₀…C₅                                                            :207-226 */
       TB_MATERIAL_HUMPHREY_STRUMPF_YIN = 6, /* C₁…C₄                                                            :235-252 */
       TB_MATERIAL_LINEAR_SPRING = 7,        /* η                                                                :261-275 */
       TB_MATERIAL_GUCCIONE_1991 = 8 };      /* C₀, Bᶠᶠ, Bˢˢ, Bⁿⁿ, Bⁿˢ, Bᶠˢ, Bᶠⁿ                                   :284-330 */
/* tb_material.reserved: the compression penalty U(I₃) (:13-87), parameters p[10] = β, p[11] = a, p[12] = b.  HO2009 with
 * TB_PENALTY_SIMPLE runs hand-derived stress / tangent routines; every other combination is differentiated on the device by
 * hyper-dual evaluation, the way the reference differentiates all of them (Tensors.hessian, materials.jl:1025-1040). */
enum { TB_PENALTY_SIMPLE = 0, TB_PENALTY_NULL = 1, TB_PENALTY_HARTMANN_NEFF_1 = 2, TB_PENALTY_HARTMANN_NEFF_2 = 3, TB_PENALTY_HARTMANN_NEFF_3 = 4 };
typedef struct tb_material {
    int32_t kind;
    int32_t reserved;           /* TB_PENALTY_* */
    double p[16];               /* p[0…8] energy parameters (HO2009 + TB_PENALTY_SIMPLE: p[8] = β), p[9] initial active tension, p[10…12] penalty β, a, b */
    double f[3], s[3], n[3];    /* ConstantCoefficient(OrthotropicMicrostructure(f, s, n)) */
    const double *fsn_field;    /* optional HOST pointer: nodal frames of an OrthotropicMicrostructureModel of FieldCoefficients
                                   (microstructure.jl:145-187), [cell][geometry node 0..7][f|s|n][3]; interpolated with the
                                   first-order shape functions, normalised and Gram–Schmidt-orthogonalised per point; NULL → f,s,n */
    int64_t fsn_field_len;      /* n_cells·8·9 */
} tb_material;

/* ionic models (src/modeling/cells/{fhn,aliev-panfilov,pcg2019}.jl) and state layouts (src/modeling/solution_variables.jl:40-68) */
enum {
    TB_CELL_FHN = 0, TB_CELL_ALIEV_PANFILOV = 1, TB_CELL_PCG2019 = 2,
    TB_CELL_TT06 = 3, /* ten Tusscher–Panfilov 2006 (epi): EXTENSION, not in the reference (BASELINE config 3 names it) */
    TB_CELL_FHN_HETEROGENEOUS = 4 /* HeterogeneousFHNModel of docs/src/literate-howto/custom-ep-cell-model.jl:8-56 with the recovery rate an affine
                                     function of the point coordinate, e(x) = e0 + g·x: parameters (a, b, c, d, e0, gx, gy, gz); needs d_x */,
    TB_CELL_ORD11 = 5 /* O'Hara–Virág–Varró–Rudy 2011 human ventricular model, 41 states: EXTENSION (SURVEY §8 f4 names it; the reference has the
                         reaction_rhs! / state_rhs! hooks only, src/modeling/cells/fhn.jl:36-60).  17 parameters: scalings of GNa, GNaL, Gto, PCa,
                         GKr, GKs, GK1, Gncx, Pnak, GKb, PNab, PCab, GpCa (1 = published), nao, cao, ko [mM], cell type (0 endo, 1 epi, 2 M) */
};
enum {
    TB_LAYOUT_SOA = 0, /* StateBlockedLayout: u[k + s·npoints]  */
    TB_LAYOUT_AOS = 1  /* PointBlockedLayout: u[k·nstates + s]  */
};

/* ------------------------------------------------------------------ errors */
const char *tb_last_error_string(void);
/* Name of the element-kernel instance the latest matrix / tangent assembly call of this thread launched, e.g. "k_patch_hex8_record<K+M,ISO,RPH20,KOFF4096>"
 * or "k_mech_points + k_mech_contract<R> + k_gather_node_rows_lds" ("" before the first such call).  Diagnostic: lets a benchmark line name the kernel
 * that ran (no reference counterpart). */
const char *tb_last_kernel_name(void);
const char *tb_version(void);
/* Revision of this interface, bumped whenever an existing entry changes what it reads or writes through its pointers OR entries are added (a
 * binding written against revision n may call anything revision n declares, so a library of revision < n must be refused up front rather than at
 * the first missing symbol).  4: tb_cgd_update writes three doubles (d_out3; revisions ≤ 3 wrote two).  5: tb_graph_*, tb_comm_exchange_begin / _end,
 * tb_cgd_iteration, tb_last_kernel_name.  6: tb_host_locality_permutation; calls that wait for the device refuse inside an open capture
 * (TB_ERR_BAD_ARG) instead of invalidating it.  A host binding compares tb_abi_revision() with the
 * TB_ABI_REVISION it was written against and refuses to run on a mismatch (julia/ThunderboltHIPBackend.jl does, in __init__) */
#define TB_ABI_REVISION 6
int tb_abi_revision(void);

/* ------------------------------------------------------------------ device (AbstractGPUDevice, src/devices.jl:3-4;
 * replaces FerriteOperators.CudaDevice as used in ext/CuThunderboltExt.jl:48-49) */
int tb_device_create(int hip_device_id, tb_device **out);
int tb_device_destroy(tb_device *dev);
/* adopt an external hipStream_t (e.g. the host framework's current stream); NULL → library-owned stream */
int tb_device_set_stream(tb_device *dev, void *hip_stream);
/* run on the legacy default (null) stream — the stream a host framework uses when it has not created one (torch.cuda.current_stream() of a
 * fresh process): kernels are then ordered with that framework's own work and collectives without events.  tb_device_set_stream(dev, NULL)
 * means "give the device its own non-blocking stream back". */
int tb_device_use_null_stream(tb_device *dev);
int tb_device_synchronize(tb_device *dev);
/* Deferred status.  By default every assembly call reads the device's status block (detJ ≤ 0 in a cell, coupling missing from the pattern) before
 * it returns — one stream synchronisation per call, the analogue of the reference's kernels that cannot throw and a host check after each
 * (src/ferrite-addons/PR883.jl:359-379).  A time loop on a fixed mesh needs that check once, not per step: with tb_device_defer_status(dev, 1) the
 * assembly calls only enqueue their kernels and return TB_OK, the flags stay raised in device memory (they are sticky), and
 * tb_device_poll_status(dev) synchronises, reports the first error raised since the last poll (TB_ERR_NEG_DETJ / TB_ERR_PATTERN, cell in
 * tb_last_error_string) and clears the block.  The solvers' own reads (CG convergence, local Newton failures) are not affected. */
int tb_device_defer_status(tb_device *dev, int on);
int tb_device_poll_status(tb_device *dev);
int tb_device_info(tb_device *dev, char *name, size_t name_len, int *n_cu, size_t *hbm_bytes);

/* vectors / matrices storage (create_system_vector / create_system_matrix, ext/CuThunderboltExt.jl:126-146) */
int tb_malloc(tb_device *dev, size_t bytes, void **d_ptr);
int tb_free(tb_device *dev, void *d_ptr);
int tb_memcpy_h2d(tb_device *dev, void *d_dst, const void *src, size_t bytes); /* synchronous */
int tb_memcpy_d2h(tb_device *dev, void *dst, const void *d_src, size_t bytes); /* synchronous */
int tb_memcpy_d2d(tb_device *dev, void *d_dst, const void *d_src, size_t bytes);
int tb_memset(tb_device *dev, void *d_ptr, int byte, size_t bytes);

/* HIP events on the device's stream (for hosts that time kernels, cf. TimerOutputs sections of
 * src/solver/time/euler.jl:85-94) */
int tb_event_create(tb_device *dev, void **event);
int tb_event_record(tb_device *dev, void *event);
int tb_event_elapsed_ms(void *start, void *stop, float *ms); /* synchronises on `stop` */
int tb_event_destroy(void *event);

/* ------------------------------------------------------------------ mesh + dof table
 * Grid + DofHandler of ONE SubDofHandler with ONE field (src/utils.jl:52-56 celldofsview,
 * dh.cell_dofs / dh.cell_dofs_offset).  The host supplies Ferrite's own node coordinates, cell
 * connectivity and dof table, so DoF indexing is inherited bit-exactly, never re-derived.
 *   xyz        n_nodes×3 (AoS);  conn  n_cells×nverts;  cell_dofs  n_cells×(nbasis·ncomp)
 *   field_kind TB_QUAD4 | TB_HEX8 | TB_TET4 | TB_HEX27 (must match geom_kind's shape); ncomp 1 (scalar) or 3.
 *   TB_QUAD4: two-dimensional meshes (the reference's GPU tests and the spiral-wave tutorial run on generate_grid(Quadrilateral, …)):
 *   coordinates are passed as n×3 with z = 0 and diffusion tensors as 3×3 with the 2×2 tensor in the upper-left block. */
int tb_mesh_create(tb_device *dev, int geom_kind, int64_t n_nodes, const double *xyz, int64_t n_cells,
                   const int32_t *conn, int field_kind, int ncomp, const int32_t *cell_dofs, int64_t ndofs,
                   int index_base, tb_mesh **out);
int tb_mesh_destroy(tb_mesh *mesh);
int64_t tb_mesh_ncells(const tb_mesh *mesh);
int64_t tb_mesh_ndofs(const tb_mesh *mesh);

/* ------------------------------------------------------------------ sparsity pattern
 * CSR pattern of allocate_matrix(dh) after transposition (src/solver/interface.jl:162-168),
 * shared by every operator of one DofHandler (src/solver/time/euler.jl:110-116, newmark.jl:105-110).
 * Builds the cell→nz scatter map on device.  Returns TB_ERR_PATTERN if a coupling is missing. */
int tb_pattern_create(tb_mesh *mesh, int64_t n_rows, const int64_t *rowptr, const int32_t *colidx,
                      int index_base, tb_pattern **out);
int tb_pattern_destroy(tb_pattern *pat);
int64_t tb_pattern_nnz(const tb_pattern *pat);
/* device copies (0-based) so the host can wrap them into its own device CSR type */
const int64_t *tb_pattern_rowptr_device(const tb_pattern *pat);
const int32_t *tb_pattern_colidx_device(const tb_pattern *pat);

/* ------------------------------------------------------------------ forms (setup_element_cache + Adapt.adapt_structure,
 * src/modeling/core/mass.jl:45-55, diffusion.jl:52-60, ext/CuThunderboltExt.jl:151-170)
 * qorder = Gauss points per direction; 0 → reference default max(2p−1,2) (src/discretization/fem.jl:52-55) */
int tb_form_create(tb_mesh *mesh, int form_kind, int qorder, const tb_coef *coef, tb_form **out);
int tb_form_destroy(tb_form *form);
/* TB_SRC_TABULATED: upload host-evaluated f(x_q,t) values, n = n_cells·nq */
int tb_form_set_table(tb_form *form, const double *values, int64_t n);

/* ------------------------------------------------------------------ assembly
 * update_operator!(op, t) for bilinear operators → fills op.A's nzval
 * (src/solver/time/euler.jl:172-176; canonical loop src/modeling/core/coordinate_systems.jl:145-171) */
int tb_assemble_matrix(tb_form *form, tb_pattern *pat, int strategy, double t, double *d_nzval);
/* The heat stage's set-up assembles the mass and the diffusion operator of one DofHandler back to back on the shared
 * sparsity pattern (update_operator!(cache.M, t); update_operator!(cache.K, t), src/solver/time/euler.jl:172-176, pattern
 * sharing :110-116): one pass over the mesh fills both nzval arrays (geometry and scatter metadata read once).  Results are
 * those of two tb_assemble_matrix calls; combinations the fused kernel does not cover fall back to exactly those two calls. */
int tb_assemble_matrix_pair(tb_form *mass, tb_form *diffusion, tb_pattern *pat, int strategy, double t, double *d_nzval_mass,
                            double *d_nzval_diffusion);
/* update_operator!(op, t) for linear operators → fills op.b (src/solver/time/euler.jl:119,176;
 * test/gpu/test_operators.jl:24-30) */
int tb_assemble_vector(tb_form *form, int strategy, double t, double *d_b);

/* ------------------------------------------------------------------ quasi-static hyperelasticity (vector field, ncomp = 3)
 * setup_element_cache(QuasiStaticModel(:u, PK1Model(...), ()), qr, sdh) (test/test_elements.jl:99-125);
 * operator creation via setup_operator(strategy, volume_integrator, dh) (src/solver/time/homotopy.jl:61-67).
 * qorder 0 → max(2p−1, 2) (src/discretization/fem.jl:52-55): 2 for Q1, 3 for Q2. */
int tb_hyperelastic_create(tb_mesh *mesh, int qorder, const tb_material *material, tb_form **out);
/* residual!(op, residual, u, p) (src/solver/nonlinear/newton_raphson.jl:234): d_r overwritten */
int tb_residual(tb_form *form, int strategy, const double *d_u, double t, double *d_r);
/* update_linearization!(op, residual, u, p) / update_linearization!(op, u, p) (newton_raphson.jl:238,
 * src/solver/time/newmark.jl:112-137): fills J's nzval, and the residual when d_r != NULL.  Dirichlet
 * elimination is NOT part of it (applied afterwards on the host: src/solver/nonlinear/nlsolve_common.jl:12-26). */
int tb_linearize(tb_form *form, tb_pattern *pat, int strategy, const double *d_u, double t, double *d_nzval, double *d_r);
/* host evaluation of the device material routine (Ψ, P = ∂Ψ/∂F, 𝔸 = ∂²Ψ/∂F²; row-major F[3i+j], A[9(3i+j)+3k+l]) */
/* Active stress (ActiveStressModel, src/modeling/solid/materials.jl:1200-1266, with SimpleActiveStress, src/modeling/solid/active.jl:100-113,
 * over a steady-state sarcomere model driven by a calcium field, src/modeling/solid/contraction.jl:103-105,166-175):
 * P += Ta·(F·f₀)⊗f₀/‖F·f₀‖ with its consistent tangent, Ta(x_q) = tension · Σₐ Mₐ(ξ_q)·state[cell][a] (first-order nodal data
 * per cell, n_cells×8 host values copied to the device) or Ta = tension when state_field is NULL.  tb_material.p[9] is the
 * initial uniform tension.  Call again whenever the calcium transient advances (host values → device, O(cells)). */
int tb_hyperelastic_set_active_tension(tb_form *form, double tension, const double *state_field, int64_t len);
/* Hill-type frameworks (src/modeling/solid/materials.jl:1042-1190): W(F) = W_passive(F) + [𝓝] W_active(F·Fᵃ⁻¹), Fᵃ from an active
 * deformation gradient model (src/modeling/solid/active.jl:23-96) and the steady-state stretch λᵃ of a calcium-driven sarcomere model
 * (src/modeling/solid/contraction.jl:166-175,302-320).  framework: generalized (no 𝓝) or extended (𝓝 = calcium-driven state).  The
 * active spring is ActiveMaterialAdapter(energy) — active_energy = TB_MATERIAL_*, active_p = 9 energy + 3 penalty parameters — or
 * SimpleActiveSpring (TB_ACTIVE_SIMPLE_SPRING, active_p[0] = aᶠ).  The calcium state reaches the kernels through
 * tb_hyperelastic_set_active_tension (tension = uniform Ca, or 1 with a nodal Ca field).  Differentiated on the device like the
 * other energies. */
enum { TB_HILL_NONE = 0, TB_HILL_GENERALIZED = 1, TB_HILL_EXTENDED = 2 };
enum { TB_ACTIVE_SIMPLE_SPRING = 100 };
enum { TB_ADG_GMK = 0, TB_ADG_GMK_INCOMPRESSIBLE = 1, TB_ADG_RLRSQ = 2 };
enum { TB_SARCOMERE_PELCE_SUN_LANGEVELD_1995 = 0 /* β, λᵃₘₐₓ */, TB_SARCOMERE_CONSTANT_STRETCH = 1 /* λ */ };
typedef struct {
    int32_t framework, active_energy, active_penalty, adg_kind, sarcomere_kind;
    double active_p[12];   /* 9 energy + 3 penalty parameters of the active spring */
    double sheetlet_part;  /* RLRSQ */
    double sarcomere_p[2];
} tb_hill;
int tb_hyperelastic_set_hill(tb_form *form, const tb_hill *hill); /* NULL or framework = TB_HILL_NONE: plain PK1Model */
/* host evaluation of the whole constitutive law at one point (the code the kernels run): activation = Ta or the calcium state */
int tb_host_material_eval_hill(const tb_material *material, const tb_hill *hill, double activation, const double *F, double *psi, double *P, double *A);
/* PrestressedMechanicalModel(inner_model, prestress_field) (src/modeling/solid/materials.jl:781-900) with a constant field:
 * P(F) = Pᵉ(F·F₀⁻¹)·F₀⁻ᵀ, ∂P/∂F_ijkl = 𝔸ᵉ_imkn F₀⁻¹_jm F₀⁻¹_ln.  F0inv: the nine entries of F₀⁻¹ (what the reference's prestress_field
 * evaluates to), row-major; NULL removes the prestress.  Evaluated through the device AD path (the product F·F₀⁻¹ is differentiated). */
int tb_hyperelastic_set_prestress(tb_form *form, const double *F0inv);

/* Subdomains: the reference integrates one material per SubDofHandler / cellset (QuasiStaticModel per subdomain,
 * test/integration/test_solid_mechanics.jl:96-140; NonlinearMultiDomainIntegrator, src/modeling/core/multi-integrator.jl).  A form with a
 * cellset integrates over those cells only; with accumulate != 0 tb_linearize / tb_residual add into their outputs instead of
 * overwriting them, so a multi-domain operator is: first form (overwrite), further forms (accumulate), facet forms (always accumulate).
 * Subdomain and accumulating forms run with TB_STRATEGY_PER_COLOR (own colouring of the subset) or TB_STRATEGY_ATOMIC. */
int tb_form_set_cellset(tb_form *form, const int32_t *cells, int64_t n_cells, int index_base);
int tb_form_clear_cellset(tb_form *form);
int tb_form_set_accumulate(tb_form *form, int accumulate);

/* Condensed internal variables: ActiveStressModel over a sarcomere model with state (RDQ20MF) whose evolution is solved per quadrature
 * point inside the assembly — QuasiStaticCondensedElementCache + solve_local_constraint (src/modeling/solid/elements.jl:411-612,
 * src/modeling/solid/materials.jl:472-502,1403-1632), rate-free local problem (AsRateIndependent: dλ/dt = 0).  With condensation set,
 * tb_linearize / tb_residual first solve (Q − Q_known)/Δt = rhs(Q, λ(F), 0, Ca) at every quadrature point (λ = ‖F f₀‖; Ca through
 * tb_hyperelastic_set_active_tension: scale · nodal field or scale) — writing Q back into d_state, which also supplies the initial guess —
 * then assemble with P = ∂Ψ/∂F + Tmax (Q₁₈+Q₂₀) fso(λ) (F f₀)⊗f₀/λ and the tangent ∂P/∂F|_Q + ∂P/∂Q · dQ/dF (corrector).
 * State arrays: n_states × n_points on the device, point-fastest, point = cell · n_qp + q (tb_hyperelastic_n_quadrature_points).
 * tb_hyperelastic_local_solve_report: failures of the last assembly (the reference's check_local_solve_convergence; a step with failed
 * points must be rejected by the caller).  sarcomere_model < 0 switches condensation off. */
int tb_hyperelastic_set_condensation(tb_form *form, int sarcomere_model, const double *params, int n_params, double tmax, double local_tol,
                                     int local_max_iters);
int tb_hyperelastic_n_quadrature_points(tb_form *form, int64_t *n_points);
int tb_hyperelastic_set_internal_state(tb_form *form, double *d_state, const double *d_state_known, double dt);
/* Rate-coupled local problem dₜQ = L(F, dₜF, Q) (the unwrapped RDQ20MFModel; QuasiStaticCondensedDAEElementCache, elements.jl:382-400;
 * solve_local_constraint materials.jl:1664-1750): dλ/dt = ∂λ/∂F : Ḟ with the backward-Euler rate Ḟ = (∇u − ∇u_prev)/Δt.  d_u_prev: the
 * accepted displacement of the previous step (device, n_dofs); NULL returns to the rate-free problem (AsRateIndependent).  The tangent
 * gains ∂P/∂Ḟ/Δt and the non-symmetric term ∂P/∂Q · ∂Q/∂λ̇ ⊗ (∂²λ/∂F² : Ḟ): use a general linear solver (tb_gmres_solve). */
int tb_hyperelastic_set_previous_solution(tb_form *form, const double *d_u_prev);
int tb_hyperelastic_local_solve_report(tb_form *form, int64_t *n_failed, int32_t *status_host, int64_t len);

/* Weak boundary conditions of a quasi-static problem (src/modeling/core/weak_boundary_conditions.jl): RobinBC
 * Ψ = α u·u (:102-198), NormalSpringBC Ψ = ½ kₛ (u·N)² (:200-300), ConstantPressureBC follower load p·J·F⁻ᵀ·n₀ with its
 * consistent tangent (:419-515).  `facets` lists n_facets pairs (cell, local facet) — Ferrite's FacetIndex, local facets of
 * the hexahedron numbered as Ferrite.reference_facets(RefHexahedron); `facet_qpoints` = Gauss points per direction on the
 * facet (the reference uses the interpolation order, src/discretization/fem.jl:80-90; 0 selects that).
 * tb_facet_assemble ADDS to d_nzval / d_r (either may be NULL): the reference accumulates surface terms into the same
 * Kₑ / rₑ as the volume term (call it after tb_linearize / tb_residual).  Vector field on hexahedra only. */
enum { TB_BC_ROBIN = 0, TB_BC_NORMAL_SPRING = 1, TB_BC_PRESSURE = 2,
       TB_BC_BENDING_SPRING = 3, /* BendingSpringBC: energy ½ kᵇ |F⁻ᵀN − N|² (:47-57, :301-415) */
       TB_BC_PRESSURE_FIELD = 4  /* PressureFieldBC: p(x) = param · first-order nodal data per cell (tb_facet_form_set_field; NULL → param) */ };
/* nodal data of a PressureFieldBC: n_cells×8 host values (per cell and geometry node), copied to the device */
int tb_facet_form_set_field(tb_form *form, const double *field, int64_t len);
int tb_facet_form_create(tb_mesh *mesh, int bc_kind, double param, int facet_qpoints, const int32_t *facets, int64_t n_facets,
                         int index_base, tb_form **out);
/* new value of the boundary condition's parameter (α, kₛ, kᵇ, p, or the scale of the nodal pressure field): time-dependent loads such as
 * the reference's ramped PressureFieldBC (test/integration/test_solid_mechanics.jl:571-590) set it before every assembly */
int tb_facet_form_set_param(tb_form *form, double param);
int tb_facet_assemble(tb_form *form, tb_pattern *pat, const double *d_u, double t, double *d_nzval, double *d_r);
int tb_host_material_eval(const tb_material *material, const double *F, double *psi, double *P, double *A);

/* ------------------------------------------------------------------ pointwise sarcomere dynamics
 * Sarcomere models with internal state (src/modeling/solid/contraction.jl:337-632).  TB_SARCOMERE_RDQ20MF: 20 states per point
 * (16 regulatory-unit occupancies, flat index (TL−1) + 2(TC−1) + 4(TR−1) + 8(CC−1); 4 cross-bridge moments), default initial state
 * (1, 0, …, 0) (default_initial_state!, :371-375); parameters in the field order of RDQ20MFModel (:337-369; 17 values, the last is εᵛ).
 * tb_sarcomere_step: the StandaloneSarcomereModel protocol (:150-163) — du = sarcomere_rhs!(u, λ, dλ/dt, Ca) — advanced by
 * `substeps` forward-Euler steps of dt with the inputs held.  State: n_states × n_points, point-fastest.  Inputs per point
 * (device arrays) or, where the pointer is NULL, the scalar argument.  rate_independent ≠ 0 evaluates at dλ/dt = 0 (AsRateIndependent,
 * :120-148).  Optional outputs after the step: compute_active_tension / compute_active_stiffness (:616-622) per point.
 * tb_host_sarcomere_eval: one evaluation of the same inline code on the host (rhs, tension, stiffness; any output may be NULL). */
enum { TB_SARCOMERE_RDQ20MF = 2 };
int tb_sarcomere_model_info(int model, int *n_states, int *n_params);
int tb_sarcomere_step(tb_device *dev, int model, const double *params, int n_params, double *d_state, int64_t n_points,
                      const double *d_stretch, const double *d_velocity, const double *d_calcium, double stretch, double velocity,
                      double calcium, double t, double dt, int substeps, int rate_independent, double *d_tension, double *d_stiffness);
int tb_host_sarcomere_eval(int model, const double *params, int n_params, const double *state, double stretch, double velocity,
                           double calcium, double *dstate, double *tension, double *stiffness);
/* The local problem of the condensed mechanics, pointwise (solve_internal_timestep + corrector, src/modeling/solid/materials.jl:1403-1568,
 * rate-free form :1575-1632, rate-coupled :1664-1750): backward Euler (Q − Q_known)/Δt = rhs(Q, λ, dλ/dt, Ca) by Newton (initial guess: d_state; `tol` on ‖residual‖₂,
 * checked like the reference: the update is applied, then the pre-update norm decides) and, if d_dstate_dstretch != NULL, the corrector
 * dQ/dλ = J⁻¹ ∂rhs/∂λ at the solution (and dQ/d(dλ/dt) into d_dstate_dvelocity, the second corrector of the rate-coupled form; a
 * velocity of 0 with d_velocity = NULL is the rate-free problem).  Per-point status (TB_LOCAL_*; the reference's LocalSolveReport retcodes) and the number of failed
 * points (n_failed != NULL synchronises).  Defaults of the reference's GenericLocalNonlinearSolver: tol 1e-4, max_iters 10. */
enum { TB_LOCAL_SUCCESS = 0, TB_LOCAL_LINEAR_SOLVE_FAILED = 1, TB_LOCAL_MAX_ITERS = 2, TB_LOCAL_CONVERGENCE_FAILURE = 3, TB_LOCAL_INFEASIBLE = 4 };
int tb_sarcomere_implicit_step(tb_device *dev, int model, const double *params, int n_params, double *d_state, const double *d_state_known,
                               int64_t n_points, const double *d_stretch, const double *d_velocity, const double *d_calcium, double stretch,
                               double velocity, double calcium, double dt, double tol, int max_iters, double *d_dstate_dstretch,
                               double *d_dstate_dvelocity, int32_t *d_status, int64_t *n_failed);
/* ∂rhs/∂state (row-major 20×20), ∂rhs/∂λ, ∂rhs/∂(dλ/dt) and rhs at one point: the hand-derived linearisation the kernels use
 * (analytic != 0) or forward-mode differentiation of the right-hand side (analytic == 0) — host-side cross-check */
int tb_host_sarcomere_derivatives(int model, const double *params, int n_params, const double *state, double stretch, double velocity, double calcium,
                                  int analytic, double *drhs_dstate, double *drhs_dstretch, double *drhs_dvelocity, double *rhs);
int tb_host_sarcomere_local_solve(int model, const double *params, int n_params, double *state, const double *state_known, double stretch,
                                  double velocity, double calcium, double dt, double tol, int max_iters, double *dstate_dstretch,
                                  double *dstate_dvelocity, int *status, int *iters, double *resnorm);

/* ------------------------------------------------------------------ pointwise reaction step
 * _pointwise_step_outer_kernel!(f, t, Δt, cache, ::DeviceVector) (src/solver/time/partitioned_solver.jl:38-52,
 * ext/CuThunderboltExt.jl:103-124).  substeps <= 1: ForwardEulerCellSolver (:80-99); substeps > 1:
 * AdaptiveForwardEulerSubstepper with reaction_threshold (:196-234).  d_du (dumat) may be NULL when the
 * caller does not need the rates (the RTC controller does: src/solver/time/rtc.jl:64-73).
 * Returns TB_OK (the reference kernels always return true; NaNs are the host's business). */
int tb_reaction_step(tb_device *dev, int model, const double *params, int n_params, double *d_u, double *d_du,
                     int64_t n_points, int n_states, int layout, double t, double dt, int substeps,
                     double threshold);
/* The same step with the per-point coordinate the reference hands to cell_rhs!(du, u, x, t, p): x = getcoordinate(cache, i)
 * (src/solver/time/partitioned_solver.jl:88-92; cache field `xs`, :63-77; Vec{sdim, Float32}, src/modeling/core/coordinate_systems.jl:43-49).
 * d_x: n_points × sdim Float32 values, point-major, or NULL ("x === nothing"); sdim 1…3.  Models that read x (TB_CELL_FHN_HETEROGENEOUS)
 * return TB_ERR_BAD_ARG without it; the others ignore it. */
int tb_reaction_step_x(tb_device *dev, int model, const double *params, int n_params, double *d_u, double *d_du,
                       int64_t n_points, int n_states, int layout, const float *d_x, int sdim, double t, double dt, int substeps,
                       double threshold);
/* Same step with the reaction tangent fused in: *rmax receives max over points of the φₘ component of the last right-hand
 * side evaluated at each point — what ReactionTangentController reads from `dumat` after the step (src/solver/time/rtc.jl:55-73)
 * — without `du` having to be written (d_du may be NULL): 16 instead of 24 bytes per DoF-update. */
int tb_reaction_step_rtc(tb_device *dev, int model, const double *params, int n_params, double *d_u, double *d_du,
                         int64_t n_points, int n_states, int layout, double t, double dt, int substeps, double threshold,
                         double *rmax);
/* Rush–Larsen step (SURVEY §8 f4 — the reference carries only the reaction_rhs!/state_rhs! hooks for it, src/modeling/cells/fhn.jl:36-60):
 * Hodgkin–Huxley-type gates are advanced with the exact solution of their linear ODE for frozen φₘ, every other state by forward
 * Euler.  Lifts the fast-gate stability limit of forward Euler (TT06: Δt = 0.02 ms in one evaluation instead of twenty sub-steps).
 * TB_CELL_TT06, TB_CELL_ORD11 and TB_CELL_PCG2019 (whose six gates relax as (g∞ − g)/τ_g, src/modeling/cells/pcg2019.jl:96-118); other models return
 * TB_ERR_UNSUPPORTED. */
int tb_reaction_step_rl(tb_device *dev, int model, const double *params, int n_params, double *d_u, int64_t n_points, int n_states,
                        int layout, double t, double dt);
int tb_cell_model_info(int model, int *n_states, int *n_params, int *phi_index);
int tb_cell_model_defaults(int model, double *params, double *u0);

/* ------------------------------------------------------------------ heat-step algebra
 * Anz = Mnz − Δt·Knz (src/solver/time/euler.jl:110-116) */
int tb_heat_matrix(tb_device *dev, int64_t nnz, const double *d_Mnz, const double *d_Knz, double dt, double *d_Anz);
/* y = α·A·x + β·y, CSR (src/utils.jl:185-231; `b = M uₙ₋₁`, src/solver/time/euler.jl:85) */
int tb_spmv_csr(tb_pattern *pat, const double *d_nzval, const double *d_x, double alpha, double beta, double *d_y);
/* Jacobi-preconditioned CG for the heat step A uₙ = b, A = M − Δt·K SPD (src/solver/time/euler.jl:94-100; the tutorials
 * configure KrylovJL_CG(atol = 1e-6, rtol = 1e-5)).  d_x holds the initial guess (uₙ₋₁) and the solution.
 * Stops when ‖r‖₂ ≤ atol + rtol·‖r₀‖₂ or after maxiter iterations; reports iterations and the final ‖r‖₂.
 * jacobi: 0 = no preconditioner, 1 = Jacobi (D⁻¹ read from d_Anz), TB_JACOBI_REUSE = Jacobi with the D⁻¹ extracted at the previous Jacobi solve
 * with this same d_Anz array, the caller vouching that its values are unchanged (the time loop solves with one matrix step after step); if another
 * array was solved with on this pattern in between, D⁻¹ is extracted again. */
enum { TB_JACOBI_REUSE = 2 };
/* the stopping threshold atol + rtol·‖r₀‖₂ of the latest tb_cg_solve / tb_cg_solve_from_residual / tb_pcg_solve / tb_gmres_solve on this pattern:
 * a solve has converged iff its reported resnorm ≤ this value (the reference fails a step whose linear solve ran into MaxIters,
 * src/solver/nonlinear/newton_raphson.jl: `solve_succeeded || return false`) */
int tb_solver_last_tolerance(const tb_pattern *pat, double *tol);
int tb_cg_solve(tb_pattern *pat, const double *d_Anz, const double *d_b, double *d_x, double rtol, double atol, int maxiter,
                int jacobi, int *iters, double *resnorm);
/* The same solve started from a known initial residual: d_r0 = b − A·x₀ supplied by the caller, so neither b nor the product A·x₀ is
 * formed.  In the backward-Euler heat step (euler.jl:71-101: A = M − Δt·K, b = M·uₙ₋₁, initial guess uₙ₋₁) the residual is
 * r₀ = Δt·K·uₙ₋₁ (+ source): one SpMV with K replaces the two with M and A. */
int tb_cg_solve_from_residual(tb_pattern *pat, const double *d_Anz, const double *d_r0, double *d_x, double rtol, double atol, int maxiter,
                              int jacobi, int *iters, double *resnorm);
/* Preconditioned CG with a choice of preconditioner.  TB_PRECOND_L1GS: ℓ₁ Gauss–Seidel, symmetric sweep, partitions of `partsize`
 * consecutive rows (the preconditioner the reference documents for its Krylov solves — Thunderbolt.Preconditioners.L1GSPrecBuilder /
 * SymmetricSweep, docs/src/api-reference/solver.md:13-22; Baker–Falgout–Kolev–Yang 2011): M = (D̃ + L_p) D̃⁻¹ (D̃ + U_p) with
 * D̃_ii = a_ii + Σ_{j outside the partition} |a_ij|.  tb_l1gs_apply: one application z = M⁻¹ r (forward or symmetric sweep). */
enum { TB_PRECOND_NONE = 0, TB_PRECOND_JACOBI = 1, TB_PRECOND_L1GS = 2,
       TB_PRECOND_CHEBYSHEV = 3 /* M⁻¹ = p_m(D⁻¹A)·D⁻¹, Chebyshev polynomial of degree m = `partsize` on [λmax/(1.8 m²), λmax] of D⁻¹A (the smoother of the
                                   reference's multigrid extension, src/solver/linear/multigrid.jl:28-33, used as a preconditioner of its own): SpMVs and one
                                   fused vector kernel per degree, no inner products; λmax by 24 Lanczos steps capped by the Gershgorin bound */ };
enum { TB_SWEEP_FORWARD = 0, TB_SWEEP_SYMMETRIC = 2 };
int tb_pcg_solve(tb_pattern *pat, const double *d_Anz, const double *d_b, double *d_x, double rtol, double atol, int maxiter, int precond,
                 int partsize, int *iters, double *resnorm);
int tb_l1gs_apply(tb_pattern *pat, const double *d_Anz, int partsize, int sweep, const double *d_r, double *d_z);
/* Restarted GMRES(restart) with right Jacobi preconditioning — the reference's default Newton inner solver
 * (LinearSolve.KrylovJL_GMRES(), src/solver/nonlinear/newton_raphson.jl:61) — for tangents that are not symmetric positive definite.
 * Same stopping test as tb_cg_solve on the true residual. */
int tb_gmres_solve(tb_pattern *pattern, const double *d_Anz, const double *d_b, double *d_x, double rtol, double atol, int maxiter, int restart,
                   int jacobi, int *iters, double *resnorm);
/* y += x (add!(b, source), src/solver/time/euler.jl:90) and max |x[i]| over a strided slice
 * (RTC reads max(dumat[:,φₘidx]), src/solver/time/rtc.jl:64-73) */
int tb_axpy(tb_device *dev, int64_t n, double a, const double *d_x, double *d_y);
int tb_absmax(tb_device *dev, int64_t n, const double *d_x, int64_t stride, double *result);
/* x·y (norms of residuals / increments in the Newton loop, src/solver/nonlinear/newton_raphson.jl:246-290) */
int tb_dot(tb_device *dev, int64_t n, const double *d_x, const double *d_y, double *result);
/* Building blocks of a Jacobi-CG on sub-domain (interface-unassembled) matrices spread over several devices — new work, the reference is
 * shared-memory only (README.md:7); its single-device counterpart is the CG of the heat stage (src/solver/time/euler.jl:94-100).  d_w weights a
 * dof held by k ranks with 1/k (NULL: 1); every scalar lives in caller-owned device memory, so the caller sums them over the ranks (RCCL
 * all-reduce) and no kernel waits for the host:
 *   tb_cgd_dot        *d_out      += Σ w·a·b
 *   tb_cgd_update     α = *d_rz / *d_pAp;  x += α p;  r −= α Ap;  d_out3[0] += Σ w·r·(D⁻¹r);  d_out3[1] += Σ w·r·r;
 *                     d_out3[2] (THREE doubles since ABI revision 4 — TB_ABI_REVISION; a two-double buffer is out of bounds — the caller zeroes it once) is set to pᵀAp when pᵀAp ≤ 0 with r·z ≠ 0 — breakdown / indefinite
 *                     operator — and never cleared: the host reads it together with ‖r‖² (the step itself is then empty, α = 0)
 *   tb_cgd_direction  β = *d_rz_new / *d_rz;  p = D⁻¹ r + β p                                   (d_dinv NULL: no preconditioner) */
int tb_cgd_dot(tb_device *dev, int64_t n, const double *d_w, const double *d_a, const double *d_b, double *d_out);
int tb_cgd_update(tb_device *dev, int64_t n, const double *d_w, const double *d_dinv, const double *d_p, const double *d_Ap, double *d_x, double *d_r,
                  const double *d_rz, const double *d_pAp, double *d_out3);
int tb_cgd_direction(tb_device *dev, int64_t n, const double *d_dinv, const double *d_r, double *d_p, const double *d_rz, const double *d_rz_new);
/* end of an iteration on the scalar block S = {rz, pAp, rz_new, rr, flag, rr of the last finished iteration} (SIX doubles) the three entries above
 * share: S[0] ← S[2], S[5] ← S[3], S[1] = S[2] = S[3] = 0 — one launch; the flag S[4] is left alone, the host reads (S[4], S[5]) */
int tb_cgd_rotate(tb_device *dev, double *d_S);
/* One whole iteration of that CG for a sub-domain without shared dofs (one rank): tb_spmv_csr_dot(pat, A, p → Ap, S[1]) → tb_cgd_update (weights = 1)
 * → tb_cgd_direction → tb_cgd_rotate on the six-double scalar block d_S, issued from ONE call (round 5: the per-call cost of an interpreted host is a
 * third of a thin slab's iteration).  Same kernels and results as the four calls. */
int tb_cgd_iteration(tb_pattern *pat, const double *d_nzval, const double *d_dinv, double *d_x, double *d_r, double *d_p, double *d_Ap, double *d_S);
/* Halo pack / unpack of the multi-GPU path — new work: the reference is shared-memory only (README.md:7); what these stand in for on one device
 * is the plain indexing of device vectors its GPU extension relies on (ext/CuThunderboltExt.jl:126-170).  Sub-domain vectors hold the dofs
 * shared with a neighbouring rank at the positions d_idx (0-based Int32, distinct within one call; both sides list the shared dofs in the same
 * order, e.g. ascending global node id):
 *   tb_gather_indexed       d_out[k] = d_vec[d_idx[k]]          pack the partial values into the send buffer
 *   tb_scatter_add_indexed  d_vec[d_idx[k]] += d_in[k]           add what the neighbour sent
 *   tb_spmv_csr_rows        d_out[k] = Σ_j A[d_rows[k], j]·x[j]   the interface rows of y = A·x straight into the send buffer, before the whole
 *                                                                product is formed: the exchange overlaps the interior SpMV
 *   tb_spmv_csr_dot         y = A·x and *d_dot += xᵀ·y           (device scalar) — Σ_ranks xᵀ·A_p·x is pᵀAp of the distributed CG: no halo needed
 * All asynchronous on the device's stream. */
int tb_gather_indexed(tb_device *dev, int64_t n, const double *d_vec, const int32_t *d_idx, double *d_out);
/* d_diag[r] = A[r, r] (0 where the pattern stores no diagonal entry): the sub-domain diagonal a distributed Jacobi preconditioner sums over the
 * interface before inverting it (single device: tb_cg_solve does this internally) */
int tb_extract_diagonal(tb_pattern *pat, const double *d_nzval, double *d_diag);
int tb_scatter_add_indexed(tb_device *dev, int64_t n, const double *d_in, const int32_t *d_idx, double *d_vec);
/* d_vec[d_idx[i]] = d_in[i] (assignment; indices of one call distinct).  The overlapped product of the distributed CG uses it to replace the
 * interface rows of the whole-domain product (formed by the stream kernel) by the rows it sent (formed by tb_spmv_csr_rows) before the received
 * rows are added: both sides of an interface then add the same two numbers and hold the same bits */
int tb_scatter_indexed(tb_device *dev, int64_t n, const double *d_in, const int32_t *d_idx, double *d_vec);
/* ---- RCCL behind the boundary (round 4; new work: the reference is shared-memory only, README.md:7) --------------------------------------------------
 * With the pack / unpack entries above a host still needed GPU-aware message passing of its own for N > 1.  These entries put the exchange itself behind the
 * ABI: one process per GPU, rank 0 calls tb_comm_unique_id and carries the TB_COMM_ID_BYTES bytes to the other ranks by whatever it has (a file, a socket,
 * MPI, Distributed.jl), every rank calls tb_comm_create (ncclCommInitRank on its device), and then
 *   tb_comm_exchange   for k < n_peers: send counts[k] doubles from d_send[k] to rank peers[k] and receive counts[k] doubles from it into d_recv[k] — ONE
 *                      grouped call (ncclGroupStart … ncclSend / ncclRecv … ncclGroupEnd) on the device's stream; a rank may name itself as a peer
 *   tb_comm_allreduce  in place over all ranks, TB_REDUCE_SUM or TB_REDUCE_MAX (the two scalar reductions of a CG iteration, timings)
 *   tb_comm_exchange_begin / tb_comm_exchange_end  (round 5) the same exchange on a queue of the communicator's own: begin orders it behind what the
 *                      device's stream holds so far (the packed send buffers) and returns; kernels launched on the device's stream between begin and end run
 *                      BESIDE the transfer; end makes the device's stream wait for it (the received values are then visible to what follows).  One
 *                      exchange in flight per communicator (TB_ERR_BAD_ARG otherwise); end without begin is a no-op.
 * All return when the work is enqueued; kernels launched behind tb_comm_exchange / tb_comm_allreduce / tb_comm_exchange_end on the device's stream are
 * ordered after it.  RCCL is opened at run time (the copy the process already holds, else librccl.so of the system; TB_RCCL_LIBRARY overrides): a
 * single-GPU user never loads it.
 * Sequence of a halo sum: tb_gather_indexed → tb_comm_exchange → tb_scatter_add_indexed; of the overlapped product: tb_spmv_csr_rows →
 * tb_comm_exchange_begin → tb_spmv_csr_dot → tb_comm_exchange_end → tb_scatter_indexed + tb_scatter_add_indexed → tb_comm_allreduce(pᵀAp). */
typedef struct tb_comm tb_comm;
#define TB_COMM_ID_BYTES 128
enum { TB_REDUCE_SUM = 0, TB_REDUCE_MAX = 1 };
int tb_comm_unique_id(void *id128);
int tb_comm_create(tb_device *dev, const void *id128, int rank, int world_size, tb_comm **out);
int tb_comm_destroy(tb_comm *comm);
int tb_comm_rank_size(tb_comm *comm, int *rank, int *size);
int tb_comm_exchange(tb_comm *comm, int n_peers, const int32_t *peers, const int64_t *counts, const double *const *d_send, double *const *d_recv);
int tb_comm_allreduce(tb_comm *comm, double *d_buf, int64_t n, int op);
int tb_comm_exchange_begin(tb_comm *comm, int n_peers, const int32_t *peers, const int64_t *counts, const double *const *d_send, double *const *d_recv);
int tb_comm_exchange_end(tb_comm *comm);
/* ---- HIP graphs behind the boundary (round 5; no reference counterpart: the reference's time loops are host loops, src/solver/time/euler.jl:71-101) ----
 * A time loop on a small sub-domain pays more for its launches than for its kernels (the 27-layer slab of an 8-GPU run: 0.41 ms of kernels in a
 * 0.47 ms step, 0.095 in a 0.136 ms CG iteration).  Between tb_graph_begin and tb_graph_end every enqueue-only call on the device — assembly into
 * fixed arrays, tb_reaction_step*, tb_spmv_csr*, tb_cgd_*, tb_heat_matrix, tb_axpy, tb_gather / tb_scatter_* — is CAPTURED instead of run;
 * tb_graph_launch replays the whole sequence with one launch.  Rules while a capture is open: the status is deferred (tb_device_defer_status
 * semantics; tb_device_poll_status after a launch reads what the replayed kernels raised); nothing that reads back to the host may be called
 * (tb_memcpy_d2h, tb_dot, the solvers' convergence looks, tb_reaction_step_rtc) — tb_graph_end then fails with TB_ERR_HIP; every plan the calls
 * need must exist already (run the sequence once, uncaptured, first).  Scalar arguments are frozen in a captured launch EXCEPT the time: forms and
 * ionic models read it from a device slot while captured, and tb_graph_launch(graph, t) sets that slot (t and cos 2πt) ahead of the replay.
 * TB_ERR_UNSUPPORTED: the HIP runtime lacks a graph call this needs (the caller falls back to plain calls).  tb_graph_node_count: nodes captured
 * (kernels, memsets), for reports. */
typedef struct tb_graph tb_graph;
int tb_graph_begin(tb_device *dev);
int tb_graph_end(tb_device *dev, tb_graph **out);
int tb_graph_launch(tb_graph *graph, double t);
int tb_graph_node_count(tb_graph *graph, int *n);
int tb_graph_destroy(tb_graph *graph);
/* Work statistics of the PATCH plan of a pattern's mesh (built on first use; no reference counterpart — the reference's strategies carry no
 * redundancy): out[0] = patches, out[1] = cell instances (a patch re-integrates the halo cells of the rows it owns: instances / cells is the
 * factor between the flops the patch kernels execute and the flops of one pass over the cells), out[2] = cells, out[3] = largest number of
 * instances in a patch, out[4] = largest number of rows in a patch, out[5] = LDS bytes of one matrix accumulator block. */
int tb_pattern_patch_stats(tb_pattern *pat, int64_t *out6);
/* Plan of the stream SpMV behind tb_spmv_csr (built with the pattern's first product): out2[0] = row signatures of the index-compressed kernel
 * (> 0: rows holding the same column offsets relative to their index share one table entry, the kernel reads 4 B per row instead of 4 B per
 * non-zero; −1: the pattern does not compress — table larger than nnz / 4 — and keeps the CSR kernel), out2[1] = entries of the signature table */
int tb_pattern_spmv_plan(tb_pattern *pat, int64_t *out2);
/* Sliced mirror of one value array for the solves that multiply a fixed matrix many times (round 4; replaces nothing in the reference, whose
 * mul!(y, A, x) of the Krylov iteration — src/solver/time/euler.jl:94-100 through LinearSolve — reads the CSC / CSR arrays directly):
 *   tb_spmv_mirror(pat, d_nzval)  copies the values of d_nzval into the pattern's mirror — slices of 64 consecutive rows, entry k of the 64 rows side by
 *                                 side, zero-padded to the slice's longest row; a second copy of the values, ≈ 0.8 ms at 2.7·10⁸ non-zeros — and binds it:
 *                                 from now on every product of this pattern with THIS pointer (tb_spmv_csr, tb_spmv_csr_dot, the products inside
 *                                 tb_cg_solve and the other solvers) reads the mirror: coalesced loads, no LDS staging, the same bits as the CSR kernel.
 *   tb_spmv_mirror(pat, NULL)     unbinds (the buffers stay for the next bind).
 * A pattern holds two mirrors (the system matrix of a solve and one more — K for the right-hand side Δt·K·uₙ₋₁ of the heat step): binding an array that is
 * already bound refreshes its mirror (and makes it the most recent), a third array takes the place of the one bound or refreshed longest ago.
 * Rewriting a bound array through the boundary — tb_assemble_matrix / tb_assemble_matrix_pair into it, tb_apply_zero_csr on it, or as the output of
 * tb_heat_matrix / tb_axpy / tb_memcpy_h2d / tb_memcpy_d2d / tb_memset (a destination anywhere inside the array counts) — drops its binding
 * (products fall back to the CSR array), and so does tb_free of it: the binding is the address, and an allocator may hand the same address to
 * the next matrix.  What the library cannot see is the caller's own kernels: beyond the entries above the caller keeps the contract that a bound
 * array is not modified; after assembling into it, or forming M − Δt·K in it again, call tb_spmv_mirror again.  Row-subset products
 * (tb_spmv_csr_rows) and the diagonal extraction keep reading the CSR array.  TB_ERR_UNSUPPORTED for patterns without a
 * mirror: 3×3-block rows (their own kernel), rows longer than 255 entries.  (Numberings whose rows share no column-offset signatures are mirrored
 * with their offsets stored entry-major beside the values: 12 instead of 8 bytes per non-zero, still coalesced.) */
int tb_spmv_mirror(tb_pattern *pat, const double *d_nzval);
int tb_spmv_csr_rows(tb_pattern *pat, const double *d_nzval, const double *d_x, int64_t n_rows, const int32_t *d_rows, double *d_out);
int tb_spmv_csr_dot(tb_pattern *pat, const double *d_nzval, const double *d_x, double *d_y, double *d_dot);
/* apply_zero!(K, f, ch) on the device CSR matrix (Ferrite.apply_zero!; CSR method src/utils.jl:263-278; used by
 * eliminate_constraints_from_linearization! / _residual! / _increment!, src/solver/nonlinear/nlsolve_common.jl:12-26):
 * d_prescribed is one byte per dof (1 = Dirichlet dof).  Rows and columns of prescribed dofs are zeroed, their diagonal
 * entry becomes `diag` (pass tb_meandiag like Ferrite does, or 1), and f is zeroed at those dofs.  d_nzval or d_f may be
 * NULL (vector-only: apply_zero!(f, ch)). */
int tb_apply_zero_csr(tb_pattern *pat, double *d_nzval, double *d_f, const uint8_t *d_prescribed, double diag);
/* mean |diagonal| (Ferrite.meandiag) */
int tb_meandiag(tb_pattern *pat, const double *d_nzval, double *result);
/* signed maximum of a strided slice: exactly `maximum(@view dumat[:, φₘidx])` of get_reaction_tangent
 * (src/solver/time/rtc.jl:64-73 — no absolute value there).  n == 0 yields −∞. */
int tb_max(tb_device *dev, int64_t n, const double *d_x, int64_t stride, double *result);

/* ------------------------------------------------------------------ Float32 value type
 * The reference types its device path by value_type(device) and its own GPU tests run Float32 (ext/CuThunderboltExt.jl:126-127,
 * test/gpu/test_operators.jl:20-31, test/gpu/ensemble-test.jl:8-40, test/gpu/diffusion-test.jl).  These entries are the `MI355XDevice{Float32,Int32}`
 * flavour of the ones above: every caller-visible array is Float32, arithmetic stays Float64 — assembly and the Krylov solve run the Float64 kernels
 * on a scratch arena owned by the device object and round the result once; the reaction step, SpMV, A = M − Δt·K and axpy convert in registers.
 * Same argument meaning and error behaviour as their Float64 namesakes; tb_malloc / tb_memcpy_* / tb_free are type-agnostic already. */
int tb_convert_f64_to_f32(tb_device *dev, int64_t n, const double *d_in, float *d_out);
int tb_convert_f32_to_f64(tb_device *dev, int64_t n, const float *d_in, double *d_out);
int tb_assemble_matrix_f32(tb_form *form, tb_pattern *pat, int strategy, double t, float *d_nzval);
int tb_assemble_matrix_pair_f32(tb_form *mass, tb_form *diffusion, tb_pattern *pat, int strategy, double t, float *d_nzval_mass, float *d_nzval_diffusion);
int tb_assemble_vector_f32(tb_form *form, int strategy, double t, float *d_b);
/* d_x NULL: tb_reaction_step; otherwise tb_reaction_step_x.  One pass: the reaction kernels instantiated on Float32 storage (states read as
 * Float32, stepped in Float64, rounded once on the way out — bit-identical to convert / step / convert back, at half the bytes moved). */
int tb_reaction_step_f32(tb_device *dev, int model, const double *params, int n_params, float *d_u, float *d_du, int64_t n_points, int n_states, int layout,
                         const float *d_x, int sdim, double t, double dt, int substeps, double threshold);
int tb_spmv_csr_f32(tb_pattern *pat, const float *d_nzval, const float *d_x, double alpha, double beta, float *d_y);
int tb_heat_matrix_f32(tb_device *dev, int64_t nnz, const float *d_Mnz, const float *d_Knz, double dt, float *d_Anz);
int tb_axpy_f32(tb_device *dev, int64_t n, double a, const float *d_x, float *d_y);
int tb_cg_solve_f32(tb_pattern *pat, const float *d_Anz, const float *d_b, float *d_x, double rtol, double atol, int maxiter, int jacobi, int *iters, double *resnorm);

/* ------------------------------------------------------------------ host-side generators (no GPU needed)
 * Ferrite-convention synthetic inputs for benchmarks and tests: generate_grid (src/mesh/generators.jl:942),
 * close!(dh) numbering, allocate_matrix pattern.  Conventions are documented in DESIGN.md (UNPINNED
 * third-party behaviour; at run time the Julia host passes Ferrite's own arrays instead). */
int tb_host_generate_grid_hex(int nx, int ny, int nz, const double *left, const double *right,
                              double *xyz, int32_t *conn);
/* generate_grid(Quadrilateral, (nx, ny), left, right): nodes x-fastest, xyz n×3 with z = 0, counter-clockwise cells */
int tb_host_generate_grid_quad(int nx, int ny, const double *left, const double *right, double *xyz, int32_t *conn);
int tb_host_perturb_nodes(int nx, int ny, int nz, double amplitude_rel, double *xyz);
int64_t tb_host_close_dofs(int field_kind, int ncomp, int64_t n_cells, int64_t n_nodes, const int32_t *conn,
                           int32_t *cell_dofs);
/* two-pass: colidx == NULL → counts only (fills rowptr, returns nnz) */
int64_t tb_host_build_pattern(int64_t n_cells, int ndofs_per_cell, const int32_t *cell_dofs, int64_t ndofs,
                              int64_t *rowptr, int32_t *colidx);

/* Locality order for an arbitrarily numbered mesh (host-side; the numbering-insensitive cell loop of src/modeling/core/coordinate_systems.jl:145-171 has
 * no use for one — the device plans do: shared scatter / row signatures, cache-resident SpMV gathers).  Sweep over the per-axis cell layers the patch
 * planner cuts (exactly the (i, j, k) layers of a distorted structured grid, density-adaptive on unstructured meshes).  Outputs, each optional (NULL):
 *   cell_perm[k]  = the cell to store k-th                     (n_cells entries)  → build the Grid with cells[cell_perm]
 *   node_perm[v]  = new number of grid node v                  (n_nodes entries)  → nodes[invperm(node_perm)], cell node ids mapped through node_perm
 *   dof_perm[d]   = new number of dof d                        (ndofs entries)    → Ferrite.renumber!(dh, dof_perm) before the pattern is allocated:
 *                   first visit when the cells are traversed in cell_perm order, local dofs in cell_dofs order — what close!(dh) numbers on a grid
 *                   stored in that order, so a lattice under any numbering receives generate_grid's own numbers back.
 * conn / cell_dofs / the outputs are index_base-based (0 or 1); cell_dofs, ndofs_per_cell, ndofs are read only when dof_perm is asked for. */
int tb_host_locality_permutation(int geom_kind, int64_t n_nodes, const double *xyz, int64_t n_cells, const int32_t *conn, int ndofs_per_cell,
                                 const int32_t *cell_dofs, int64_t ndofs, int index_base, int32_t *cell_perm, int32_t *node_perm, int32_t *dof_perm);

#ifdef __cplusplus
}
#endif
#endif /* TBHIP_H */
