/* abi_driver.c — a caller of libtbhip.so that is neither Python nor Julia: plain C over include/tbhip.h, the calls a maintainer's ccall
 * layer would make, in the order the reference makes them.
 *
 *   set-up            setup_solver_cache of the heat stage: mass + diffusion operators of one DofHandler assembled on the shared pattern
 *                     (src/solver/time/euler.jl:143-176), source operator (:148-153)
 *   every time step   perform_backward_euler_step! (src/solver/time/euler.jl:71-101): A = M − Δt·K when Δt changed (:104-116), b = M·uₙ₋₁ (:85),
 *                     update_operator!(source, t + Δt); add!(b, source) (:88-90), solve A·uₙ = b from the initial guess uₙ₋₁ (:94-100);
 *                     then the cell step of the operator splitting, _pointwise_step_outer_kernel! with the forward-Euler inner kernel
 *                     (src/solver/time/partitioned_solver.jl:38-52,80-99) on the state blocked as u[k + s·npoints] (fem.jl:385-408)
 *
 * 16³ trilinear hexahedra on [0,1]³ (smoothly distorted), monodomain diffusivity κ/(Cₘχ), FitzHugh–Nagumo cells, spiral-wave-like initial
 * data (ep01_spiral-wave.jl:113-118).  Prints the final state (φₘ of every dof, hex floats) for tests/test_abi_driver.py, which repeats the
 * same steps with the CPU oracle.   build: gcc -O2 -std=c11 -I include tests/abi_driver.c -L thunderbolt.jl_amd -ltbhip -lm -o tests/abi_driver */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "tbhip.h"

#define CK(call)                                                                                     \
    do {                                                                                             \
        int rc__ = (call);                                                                           \
        if (rc__ != TB_OK) {                                                                         \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc__, tb_last_error_string());                  \
            return 1;                                                                                \
        }                                                                                            \
    } while (0)

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 16, nsteps = argc > 2 ? atoi(argv[2]) : 5;
    const double dt = 0.5, left[3] = {0, 0, 0}, right[3] = {1, 1, 1};
    const int64_t n_nodes = (int64_t)(n + 1) * (n + 1) * (n + 1), n_cells = (int64_t)n * n * n;
    double *xyz = malloc(sizeof(double) * 3 * n_nodes);
    int32_t *conn = malloc(sizeof(int32_t) * 8 * n_cells), *cell_dofs = malloc(sizeof(int32_t) * 8 * n_cells);
    CK(tb_host_generate_grid_hex(n, n, n, left, right, xyz, conn));
    CK(tb_host_perturb_nodes(n, n, n, 0.2, xyz));
    const int64_t ndofs = tb_host_close_dofs(TB_HEX8, 1, n_cells, n_nodes, conn, cell_dofs);
    int64_t *rowptr = malloc(sizeof(int64_t) * (ndofs + 1));
    const int64_t nnz = tb_host_build_pattern(n_cells, 8, cell_dofs, ndofs, rowptr, NULL);
    int32_t *colidx = malloc(sizeof(int32_t) * nnz);
    if (tb_host_build_pattern(n_cells, 8, cell_dofs, ndofs, rowptr, colidx) != nnz) return 2;

    tb_device *dev;
    tb_mesh *mesh;
    tb_pattern *pat;
    CK(tb_device_create(0, &dev));
    CK(tb_mesh_create(dev, TB_HEX8, n_nodes, xyz, n_cells, conn, TB_HEX8, 1, cell_dofs, ndofs, 0, &mesh));
    CK(tb_pattern_create(mesh, ndofs, rowptr, colidx, 0, &pat));

    /* BilinearMassIntegrator(ConstantCoefficient(1)), BilinearDiffusionIntegrator(ConductivityToDiffusivityCoefficient(κ, Cₘ, χ)), LinearIntegrator(source) */
    tb_coef cm = {0}, ck = {0}, cs = {0};
    cm.kind = TB_COEF_CONST_SCALAR; cm.p[0] = 1.0;
    ck.kind = TB_COEF_CONST_TENSOR; ck.wrap = 1; ck.Cm = 1.0; ck.chi = 2.0;
    ck.p[0] = 4.5e-3; ck.p[4] = 2.0e-3; ck.p[8] = 2.0e-3; ck.p[1] = ck.p[3] = 5.0e-4;
    cs.kind = TB_SRC_COS_EXP;
    tb_form *fm, *fk, *fs;
    CK(tb_form_create(mesh, TB_FORM_MASS, 0, &cm, &fm));
    CK(tb_form_create(mesh, TB_FORM_DIFFUSION, 0, &ck, &fk));
    CK(tb_form_create(mesh, TB_FORM_SOURCE, 0, &cs, &fs));

    double *d_M, *d_K, *d_A, *d_b, *d_src, *d_u;
    CK(tb_malloc(dev, sizeof(double) * nnz, (void **)&d_M));
    CK(tb_malloc(dev, sizeof(double) * nnz, (void **)&d_K));
    CK(tb_malloc(dev, sizeof(double) * nnz, (void **)&d_A));
    CK(tb_malloc(dev, sizeof(double) * ndofs, (void **)&d_b));
    CK(tb_malloc(dev, sizeof(double) * ndofs, (void **)&d_src));
    CK(tb_malloc(dev, sizeof(double) * 2 * ndofs, (void **)&d_u)); /* (φₘ, s) blocked by state */
    CK(tb_assemble_matrix_pair(fm, fk, pat, TB_STRATEGY_PATCH, 0.0, d_M, d_K));   /* euler.jl:172-176 */

    /* initial condition on the dofs: φₘ = 1 where x, y ≤ ½, s = 0.1 where y ≥ ½ (ep01_spiral-wave.jl:113-118) */
    double *u = calloc(2 * ndofs, sizeof(double));
    for (int64_t c = 0; c < n_cells; ++c)
        for (int a = 0; a < 8; ++a) {
            const double *x = xyz + 3 * conn[8 * c + a];
            const int32_t d = cell_dofs[8 * c + a];
            u[d] = (x[0] <= 0.5 && x[1] <= 0.5) ? 1.0 : 0.0;
            u[ndofs + d] = x[1] >= 0.5 ? 0.1 : 0.0;
        }
    CK(tb_memcpy_h2d(dev, d_u, u, sizeof(double) * 2 * ndofs));

    double fhn[6], dummy[2];
    CK(tb_cell_model_defaults(TB_CELL_FHN, fhn, dummy));
    double dt_last = -1.0, t = 0.0;
    int total_iters = 0;
    for (int step = 0; step < nsteps; ++step, t += dt) {
        if (dt != dt_last) {                                                                     /* :104-116 */
            CK(tb_heat_matrix(dev, nnz, d_M, d_K, dt, d_A)); dt_last = dt;
            CK(tb_spmv_mirror(pat, d_A));   /* A is fixed until Δt changes: the products of the solves below stream its sliced mirror (same bits) */
            CK(tb_spmv_mirror(pat, d_M));   /* … and so does b = M uₙ₋₁ (a pattern holds two mirrors) */
        }
        CK(tb_spmv_csr(pat, d_M, d_u, 1.0, 0.0, d_b));                                           /* b = M uₙ₋₁, :85 */
        CK(tb_assemble_vector(fs, TB_STRATEGY_ELEMENT, t + dt, d_src));                         /* update_operator!(source, t + Δt), :88 (EA strategy, :148-153) */
        CK(tb_axpy(dev, ndofs, 1.0, d_src, d_b));                                                /* add!(b, source), :90 */
        int iters;
        double res;
        CK(tb_cg_solve(pat, d_A, d_b, d_u, 1e-10, 1e-12, 500, 1, &iters, &res));                 /* :94-100, initial guess uₙ₋₁ = φₘ block of u */
        total_iters += iters;
        CK(tb_reaction_step(dev, TB_CELL_FHN, fhn, 6, d_u, NULL, ndofs, 2, TB_LAYOUT_SOA, t, dt, 1, 0.0)); /* partitioned_solver.jl:38-52 */
    }
    CK(tb_memcpy_d2h(dev, u, d_u, sizeof(double) * 2 * ndofs));
    printf("ndofs %lld nnz %lld steps %d cg_iters %d\n", (long long)ndofs, (long long)nnz, nsteps, total_iters);
    for (int64_t i = 0; i < 2 * ndofs; ++i) printf("%a\n", u[i]);
    tb_free(dev, d_M); tb_free(dev, d_K); tb_free(dev, d_A); tb_free(dev, d_b); tb_free(dev, d_src); tb_free(dev, d_u);
    tb_form_destroy(fm); tb_form_destroy(fk); tb_form_destroy(fs);
    tb_pattern_destroy(pat); tb_mesh_destroy(mesh); tb_device_destroy(dev);
    free(xyz); free(conn); free(cell_dofs); free(rowptr); free(colidx); free(u);
    return 0;
}
