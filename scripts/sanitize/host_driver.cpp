// AddressSanitizer / UBSan harness for the host-side code that runs without a GPU: mesh / dof / pattern generators of the
// product (tb_hostgen.cpp) cross-checked against the oracle, and the oracle's element, reaction, mechanics and facet routines
// (sequential and OpenMP).  GPU sanitizers are not available on the pool; run: bash scripts/sanitize/run.sh
#include <cstdio>
#include <cstdarg>
#include <vector>
#include <cstdint>
#include "tbhip.h"
extern "C" {
#include "tb_oracle.h"
}
namespace tb { void set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); } }
int main()
{
    for (int kind : {TB_HEX8, TB_HEX27}) {
        const int nx = 5, ny = 4, nz = 3;
        std::vector<double> xyz(3 * (nx + 1) * (ny + 1) * (nz + 1));
        std::vector<int32_t> conn(8 * nx * ny * nz);
        double le[3] = {0, 0, 0}, ri[3] = {1, 1, 1};
        if (tb_host_generate_grid_hex(nx, ny, nz, le, ri, xyz.data(), conn.data())) return 1;
        tb_host_perturb_nodes(nx, ny, nz, 0.2, xyz.data());
        for (int ncomp : {1, 3}) {
            const int nb = kind == TB_HEX27 ? 27 : 8;
            std::vector<int32_t> cd((size_t)nx * ny * nz * nb * ncomp);
            const int64_t nd = tb_host_close_dofs(kind, ncomp, nx * ny * nz, (int64_t)xyz.size() / 3, conn.data(), cd.data());
            if (nd <= 0) return 2;
            std::vector<int64_t> rp(nd + 1);
            const int64_t nnz = tb_host_build_pattern(nx * ny * nz, nb * ncomp, cd.data(), nd, rp.data(), nullptr);
            std::vector<int32_t> ci(nnz);
            if (tb_host_build_pattern(nx * ny * nz, nb * ncomp, cd.data(), nd, rp.data(), ci.data()) != nnz) return 3;
            // oracle on the same inputs
            std::vector<int32_t> cd2(cd.size());
            const int64_t nd2 = orc_close_dofs(kind, ncomp, nx * ny * nz, (int64_t)xyz.size() / 3, conn.data(), cd2.data());
            if (nd2 != nd || cd2 != cd) { fprintf(stderr, "dof mismatch\n"); return 4; }
            std::vector<int64_t> rp2(nd + 1);
            const int64_t nnz2 = orc_build_pattern(nx * ny * nz, nb * ncomp, cd.data(), nd, rp2.data(), nullptr);
            if (nnz2 != nnz) return 5;
            // oracle element routines on the same mesh (sequential and OpenMP paths)
            orc_mesh om{kind == TB_HEX27 ? ORC_HEX27 : ORC_HEX8, kind == TB_HEX27 ? 3 : 2, nx * ny * nz, (int64_t)xyz.size() / 3, xyz.data(), conn.data(), cd.data()};
            if (ncomp == 1 && kind == TB_HEX8) {
                std::vector<int32_t> color(nx * ny * nz);
                const int ncol = orc_color_cells(nx * ny * nz, nb, cd.data(), nd, color.data());
                std::vector<double> nz(nnz), b(nd);
                double one = 1.0, D[9] = {2, .3, .1, .3, 1.5, -.2, .1, -.2, 1};
                orc_coef cm{ORC_COEF_CONST_SCALAR, &one, nullptr, 1, 1, 0}, ck{ORC_COEF_CONST_TENSOR, D, nullptr, 1.3, 0.7, 1};
                for (int th : {1, 4}) {
                    if (orc_assemble_matrix(&om, 0, &cm, 0.0, rp.data(), ci.data(), nz.data(), th, color.data(), ncol)) return 10;
                    if (orc_assemble_matrix(&om, 1, &ck, 0.0, rp.data(), ci.data(), nz.data(), th, color.data(), ncol)) return 11;
                    if (orc_assemble_source(&om, ORC_SRC_COS_EXP, &one, nullptr, 0.1, b.data(), th)) return 12;
                }
                for (int model = 0; model < 4; ++model) {
                    double p[64], u0[32];
                    orc_cell_default_params(model, p);
                    orc_cell_default_state(model, p, u0);
                    const int ns = orc_cell_nstates(model);
                    std::vector<double> u((size_t)ns * 37), du(u.size());
                    for (int s = 0; s < ns; ++s) for (int k = 0; k < 37; ++k) u[s * 37 + k] = u0[s];
                    for (int th : {1, 3}) if (orc_reaction_step(model, p, u.data(), du.data(), 37, 0, 0.0, 1e-3, 4, 0.05, th)) return 13;
                }
            }
            if (ncomp == 3) {
                std::vector<double> u(nd), nz(nnz), r(nd);
                for (int64_t i = 0; i < nd; ++i) u[i] = 1e-2 * ((i * 7919) % 13 - 6) / 6.0;
                const double p[9] = {0.059, 8.023, 18.472, 16.026, 2.581, 11.120, 0.216, 11.436, 1.0}, fsn[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
                if (orc_assemble_hyperelastic(&om, p, fsn, u.data(), rp.data(), ci.data(), nz.data(), r.data(), 1, nullptr, 0)) return 14;
                const int32_t facets[4] = {0, 4, nx - 1, 2};
                for (int bc = 0; bc < 3; ++bc)
                    if (orc_assemble_facets(&om, bc, 0.3, kind == TB_HEX27 ? 2 : 1, facets, 2, u.data(), rp.data(), ci.data(), nz.data(), r.data())) return 15;
            }
            printf("kind %d ncomp %d: ndofs %lld nnz %lld ok\n", kind, ncomp, (long long)nd, (long long)nnz);
        }
    }
    // quads
    {
        std::vector<double> xyz(3 * 8 * 3); std::vector<int32_t> conn(4 * 7 * 2);
        double le[2] = {-1, -1}, ri[2] = {1, 1};
        if (tb_host_generate_grid_quad(7, 2, le, ri, xyz.data(), conn.data())) return 6;
        std::vector<int32_t> cd(conn.size());
        if (tb_host_close_dofs(TB_QUAD4, 1, 14, 24, conn.data(), cd.data()) != 24) return 7;
    }
    puts("host generators clean");
    return 0;
}
