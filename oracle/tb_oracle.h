/*
 * tb_oracle.h — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT).
 *
 * Plain-C restatement of the Thunderbolt.jl v0.0.4 hot path (per-cell FE integration + pointwise
 * ionic reaction step).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the shipped path (thunderbolt.jl_amd/csrc) never links, imports or calls it.
 *
 * PARITY STATUS: the Julia reference cannot be executed in the build container (no julia binary,
 * Ferrite.jl / FerriteOperators.jl / Tensors.jl are not vendored under /root/reference).  The
 * coefficient evaluators are pinned against the closed-form expectations of the reference's own
 * test/test_coefficients.jl (tests/golden/coefficients.json); the geometry stage is pinned by the
 * identities of the reference's "Static interpolation values" test on its distorted cells; the cell
 * models / steppers are closed forms.  Assembled K, M, b, DoF numbering and sparsity order have no
 * golden vector in the reference: for those rows this oracle is "parity unpinned" (see DESIGN.md).
 *
 * All file:line citations are relative to /root/reference.
 */
#ifndef TB_ORACLE_H
#define TB_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference cell kinds (Ferrite reference shapes, [-1,1]^d hypercubes, unit simplices) */
enum { ORC_LINE2 = 1, ORC_QUAD4 = 2, ORC_HEX8 = 3, ORC_TET4 = 4, ORC_HEX27 = 5 };

/* coefficient kinds */
enum {
    ORC_COEF_CONST_SCALAR = 0,   /* p[0]                                         coefficients.jl:101-120 */
    ORC_COEF_CONST_TENSOR = 1,   /* p[0..d*d) row-major                          coefficients.jl:101-120 */
    ORC_COEF_FIELD_SCALAR = 2,   /* field[basis + nb*cell]                       coefficients.jl:85-99   */
    ORC_COEF_SPECTRAL_CONST = 3, /* p = f,s,n (d each), lambda[d]; no orthogonalisation microstructure.jl:136-138 */
    ORC_COEF_SPECTRAL_FIELD = 4, /* field = f,s,n nodal Vec3 per cell; p = lambda[3]  microstructure.jl:176-187 */
    ORC_COEF_TRANSVERSE_CONST = 5/* p = f (d), lambda[2]                         microstructure.jl:89-92 */
};

/* closed-form source terms f(x,t) (Julia closures cannot cross a C ABI — SURVEY F10) */
enum {
    ORC_SRC_CONST = 0,             /* p[0] */
    ORC_SRC_NORM_PLUS_T = 1,       /* norm(x)+t              benchmarks/benchmarks-linear-form.jl:16-20 */
    ORC_SRC_COS_EXP = 2,           /* cos(2πt)·exp(-‖x‖²)    test/gpu/test_operators.jl:13-18 */
    ORC_SRC_TABULATED = 3          /* table[q + nq*cell] (host-evaluated closure) */
};

enum { ORC_CELL_FHN = 0, ORC_CELL_ALIEV_PANFILOV = 1, ORC_CELL_PCG2019 = 2, ORC_CELL_TT06 = 3 /* extension, not in the reference */,
       ORC_CELL_FHN_HETEROGENEOUS = 4 /* docs/src/literate-howto/custom-ep-cell-model.jl:8-56 with e(x) = e0 + g·x */,
       ORC_CELL_ORD11 = 5 /* O'Hara–Virág–Varró–Rudy 2011 human ventricular model, 41 states: extension, pinned on the paper */ };
enum { ORC_LAYOUT_SOA = 0, ORC_LAYOUT_AOS = 1 };

/* ---- FE substrate (Ferrite conventions restated; UNPINNED where SURVEY §8c says so) ---- */
int orc_elem_info(int kind, int *refdim, int *nbasis);
/* Gauss rule with `order` points per direction (hypercubes) / Ferrite's order-`order` simplex rule.
 * xi: nq*refdim, w: nq. returns nq (or <0). */
int orc_quadrature(int kind, int order, double *xi, double *w);
/* N[nb], dN[nb*refdim] at reference point xi */
int orc_shape(int kind, const double *xi, double *N, double *dN);

/* src/ferrite-addons/PR883.jl:253-263 + :367-387 — J, detJ, dNdx = dNdξ·J⁻¹ (square J only) */
int orc_mapping(int dim, int ngeo, const double *x, const double *dM, double *J, double *detJ, double *Jinv);

/* ---- coefficient evaluators at one quadrature point (dimension generic) ---- */
/* src/modeling/core/coefficients.jl:85-99 */
void orc_eval_field(int nb, int ncomp, const double *Nq, const double *data_cell, double *out);
/* src/modeling/core/coefficients.jl:279-292 */
void orc_eval_cartesian(int nb, int sdim, const double *Nq, const double *coords, double *x);
/* src/modeling/microstructure.jl:36-38,89-92,136-138 */
void orc_eval_spectral(int dim, int nvec, const double *vecs, const double *lambda, double *D);
/* src/utils.jl:131-139 (3D) and :120-124 (2D) */
void orc_orthogonalize(int dim, double *f, double *s, double *n);
/* src/modeling/core/coefficients.jl:152-162: κ/(Cₘ·χ) */
void orc_conductivity_to_diffusivity(int n, const double *kappa, double Cm, double chi, double *D);
/* src/modeling/core/coefficients.jl:519-531 */
int orc_eval_homogeneous_data_index(int ntimings, const double *timings, double t);

/* ---- mesh / dof / pattern generators (Ferrite.jl 1.6 conventions, UNPINNED) ---- */
void orc_generate_grid_hex(int nx, int ny, int nz, const double *left, const double *right,
                           double *xyz, int32_t *conn);
/* Ferrite close!(dh): first-visit numbering, cells in order, vertices→edges→faces→volume,
 * `ncomp` components interleaved per entity. kind HEX8 (order 1) or HEX27 (order 2 field on HEX8 grid).
 * cell_dofs: n_cells*ndofs_per_cell (0-based). returns ndofs. */
int64_t orc_close_dofs(int kind, int ncomp, int64_t n_cells, int64_t n_nodes, const int32_t *conn,
                       int32_t *cell_dofs);
/* allocate_matrix(dh): all intra-cell couplings, sorted columns per row. Two-pass:
 * colidx==NULL → fills rowptr only; returns nnz. */
int64_t orc_build_pattern(int64_t n_cells, int ndpc, const int32_t *cell_dofs, int64_t ndofs,
                          int64_t *rowptr, int32_t *colidx);
/* greedy colouring of the cell-conflict graph (cells sharing a dof) — semantics of
 * PerColorAssemblyStrategy; returns ncolors */
int orc_color_cells(int64_t n_cells, int ndpc, const int32_t *cell_dofs, int64_t ndofs, int32_t *color);

/* ---- element kernels ---- */
typedef struct {
    int kind;          /* geometry/field cell kind (ORC_HEX8, ORC_TET4, ...) */
    int qorder;        /* quadrature order (points per direction) */
    int64_t n_cells, n_nodes;
    const double *xyz; /* n_nodes*dim */
    const int32_t *conn;
    const int32_t *cell_dofs; /* n_cells*nb */
} orc_mesh;

typedef struct {
    int kind;
    const double *p;      /* parameters */
    const double *field;  /* per-cell nodal data or tabulated values */
    double Cm, chi;       /* applied as D/(Cm*chi) when wrap != 0  (fem.jl:413-419) */
    int wrap;
} orc_coef;

/* mass.jl:28-43 */
int orc_element_mass(const orc_mesh *m, int64_t cell, const orc_coef *rho, double t, double *Me);
/* diffusion.jl:28-50 + utils.jl:409-410 */
int orc_element_diffusion(const orc_mesh *m, int64_t cell, const orc_coef *D, double t, double *Ke);
/* analytical_coefficient.jl:80-101 */
int orc_element_source(const orc_mesh *m, int64_t cell, int src_kind, const double *p,
                       const double *table, double t, double *be);

/* ---- drivers: sequential cell loop + assemble! (coordinate_systems.jl:145-171, :203-230) ---- */
/* form: 0 = mass, 1 = diffusion. nzval overwritten. nthreads>1 → per-colour OpenMP (baseline only) */
int orc_assemble_matrix(const orc_mesh *m, int form, const orc_coef *c, double t,
                        const int64_t *rowptr, const int32_t *colidx, double *nzval,
                        int nthreads, const int32_t *color, int ncolors);
/* CPU-baseline form of the per-colour assembly (bench.py only): precomputed scatter positions, per-colour cell lists, first-touch zero fill */
typedef struct orc_assembly_plan orc_assembly_plan;
int orc_assembly_plan_create(const orc_mesh *m, const int64_t *rowptr, const int32_t *colidx, const int32_t *color, int ncolors, int nthreads,
                             orc_assembly_plan **out);
void orc_assembly_plan_destroy(orc_assembly_plan *p);
int orc_assemble_matrix_planned(const orc_assembly_plan *p, const orc_mesh *m, int form, const orc_coef *c, double *nzval, int nthreads);
int orc_assemble_source_planned(const orc_assembly_plan *p, const orc_mesh *m, int src_kind, const double *par, const double *table, double t, double *b, int nthreads);
int orc_assemble_source(const orc_mesh *m, int src_kind, const double *p, const double *table,
                        double t, double *b, int nthreads);

/* ---- reaction ---- */
/* cells/fhn.jl:21-34, cells/aliev-panfilov.jl:17-31, cells/pcg2019.jl:52-133 */
int orc_cell_nstates(int model);
int orc_cell_nparams(int model);
void orc_cell_default_params(int model, double *p);
void orc_cell_default_state(int model, const double *p, double *u0);   /* pcg2019.jl:137-152 */
void orc_cell_rhs(int model, const double *p, const double *u, double t, double *du);
/* partitioned_solver.jl:38-52 (outer), :80-99 (FE), :196-234 (adaptive). substeps<=1 → plain FE. */
int orc_reaction_step(int model, const double *p, double *u, double *du, int64_t npoints, int layout,
                      double t, double dt, int substeps, double threshold, int nthreads);

/* the same outer loop with the point coordinate handed to cell_rhs! (partitioned_solver.jl:88-92): xs = npoints × sdim Float32, or NULL */
int orc_reaction_step_x(int model, const double *p, double *u, double *du, int64_t npoints, int layout, const float *xs, int sdim,
                        double t, double dt, int substeps, double threshold, int nthreads);
int orc_reaction_step_rl(int model, const double *p, double *u, int64_t npoints, int layout, double t, double dt, int nthreads);

/* ---- heat-step algebra ---- */
/* euler.jl:110-116 */
void orc_heat_matrix(int64_t nnz, const double *Mnz, const double *Knz, double dt, double *Anz);
/* utils.jl:185-231 */
void orc_spmv_csr(int64_t nrows, const int64_t *rowptr, const int32_t *colidx, const double *nz,
                  const double *x, double alpha, double beta, double *y, int nthreads);

#ifdef __cplusplus
}
#endif
#endif

/* ---- quasi-static hyperelasticity (appended) ---- */
#ifdef __cplusplus
extern "C" {
#endif
/* material parameters of HolzapfelOgden2009Model + SimpleCompressionPenalty, src/modeling/solid/energies.jl:136-146,80-87:
 * p = a, b, af, bf, as, bs, afs, bfs, beta */
/* Ψ(F), P = ∂Ψ/∂F, A = ∂²Ψ/∂F² by forward-mode (hyper-dual) differentiation of the energy, exactly what
 * Tensors.gradient / Tensors.hessian do in src/modeling/solid/materials.jl:1025-1040.
 * F, P row-major 3×3 (F[i][j] = ∂x_i/∂X_j); A[9*(3i+j) + 3k+l] = ∂P_ij/∂F_kl; fsn = f,s,n (3 each). */
double orc_ho_energy(const double *p, const double *fsn, const double *F, double *P, double *A);
/* src/modeling/solid/elements.jl:177-225 (K + r), :227-273 (K), :275-313 (r): cell of a HEX8 (order 1) or HEX27
 * (order 2, trilinear geometry) vector field, ue = element unknowns (node-major, component-minor).
 * Ke (ndofs×ndofs) and/or re (ndofs) may be NULL; both are ACCUMULATED into. */
int orc_element_hyperelastic(const orc_mesh *m, int64_t cell, const double *p, const double *fsn,
                             const double *ue, double *Ke, double *re);
/* update_linearization!(op, residual, u, p) / residual!(op, residual, u, p) call sites
 * src/solver/nonlinear/newton_raphson.jl:234-238: zero-fill, cell loop, load_element_unknowns!
 * (elements.jl:125-132), element kernel, assemble!.  nzval and/or r may be NULL. */
/* nodal f,s,n field for the hyperelastic routines ([cell][geometry node][f|s|n][3]); NULL restores the constant frame */
void orc_set_microstructure_field(const double *field);
int orc_assemble_hyperelastic(const orc_mesh *m, const double *p, const double *fsn, const double *u,
                              const int64_t *rowptr, const int32_t *colidx, double *nzval, double *r, int nthreads,
                              const int32_t *color, int ncolors);

/* ---- weak boundary conditions on hexahedron facets (src/modeling/core/weak_boundary_conditions.jl) ----
 * kind: RobinBC Ψ = α u·u (:102-168), NormalSpringBC Ψ = ½ kₛ (u·N)² (:200-300), ConstantPressureBC follower load
 * p·J·F⁻ᵀ·n₀ with its tangent (:419-515).  FacetValues conventions are Ferrite's [UNPINNED]: reference facets of
 * RefHexahedron (1,4,3,2),(1,2,6,5),(2,3,7,6),(3,4,8,7),(1,5,8,4),(5,6,7,8); Gauss–Legendre tensor rule with `fq` points
 * per direction on the facet (FacetQuadratureRuleCollection(interpolation order), src/discretization/fem.jl:80-90);
 * dΓ·n₀ = ∂x/∂s × ∂x/∂t · w, oriented outwards.  facets: nf pairs (cell, local facet 0…5).
 * Contributions are ADDED to nzval / r (the reference accumulates them into the same Kₑ / rₑ as the volume term). */
#define ORC_BC_ROBIN 0
#define ORC_BC_NORMAL_SPRING 1
#define ORC_BC_PRESSURE 2
#define ORC_BC_BENDING_SPRING 3 /* ½ kᵇ |F⁻ᵀN − N|² (:47-57, :301-415) */
#define ORC_BC_PRESSURE_FIELD 4 /* PressureFieldBC: p = param · nodal field (:71-77, :516-632) */
void orc_set_facet_pressure_field(const double *field_per_cell_node /* n_cells×8 or NULL */);
/* any energy of src/modeling/solid/energies.jl (ids as in thunderbolt.jl_amd/csrc/tb_energy.hpp) by hyper-dual AD */
double orc_energy(int energy, int penalty, const double *p, const double *up, const double *fsn, const double *F, double *P, double *A);
void orc_set_point_activation(double a);
void orc_set_hill(int framework, int act_energy, int act_penalty, const double *act_p, int adg, double kappa, int sarc, const double *sarc_p);
void orc_set_material(int energy, int penalty, const double *p /* 9 */, const double *up /* 3: β, a, b */);
void orc_set_active_tension(double tension, const double *field_per_cell_node /* n_cells×8 or NULL */);
int orc_element_facet(const orc_mesh *m, int64_t cell, int local_facet, int kind, double param, int fq, const double *ue,
                      double *Ke /* nd×nd or NULL, accumulated */, double *re /* nd or NULL, accumulated */);
int orc_assemble_facets(const orc_mesh *m, int kind, double param, int fq, const int32_t *facets, int64_t nf, const double *u,
                        const int64_t *rowptr, const int32_t *colidx, double *nzval, double *r);

#ifdef __cplusplus
}
/* RDQ20-MF sarcomere model (contraction.jl:337-622); pinned by the reference's golden trajectory */
void orc_rdq20mf_rhs(const double *p, const double *u, double lam, double dlam, double Ca, double *du);
double orc_rdq20mf_overlap(const double *p, double lam);
double orc_rdq20mf_tension(const double *p, const double *u, double lam);
double orc_rdq20mf_stiffness(const double *p, const double *u, double lam);
void orc_rdq20mf_trajectory(const double *p, double *u, int64_t n_steps, double dt, const double *lam, const double *dlam, const double *ca,
                            const unsigned char *sample, double *out);

int orc_rdq20mf_local_solve(const double *p, double *Q, const double *Qknown, double lam, double dlam, double Ca, double dt, double tol, int max_iters,
                            double *dQdl, int *iters, double *resnorm);

/* condensed internal variable of the hyperelastic element routines (global test state; p = NULL switches it off) */
void orc_set_condensation(const double *p, double tmax, double *Q, const double *Qknown, int64_t n_points, double dt, double tol, int max_iters, int *status);

void orc_set_prestress(const double *F0inv); /* NULL: none */

void orc_set_condensation_rate(const double *u_prev); /* after orc_set_condensation; NULL → rate-free */
int orc_rdq20mf_local_solve_rate(const double *p, double *Q, const double *Qknown, double lam, double dlam, double Ca, double dt, double tol, int max_iters,
                                 double *dQdl, double *dQdv, int *iters, double *resnorm);

#endif
