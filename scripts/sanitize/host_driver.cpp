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
// the header-only device routines, compiled for the host
#define TB_HD inline
#include "tb_energy.hpp"
#include "tb_sarcomere.hpp"

static int sarcomere_and_materials()
{
    // oracle: sarcomere rhs, trajectory, local solves (rate-free and rate-coupled), against the host version of the device algebra
    const double p[17] = {1.25, 1.65, 0.18, 2.2, 2.0, 0.381, -0.571, 10.0, 12.0, 0.1, 0.013, 0.13431, 25.184, 0.032653, 0.000778, 22.894e3, 1.0e-6};
    tbk::RDQ20Params P{p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], p[9], p[10], p[11], p[12], p[13], p[14], p[15], p[16]};
    double u[20] = {1.0}, du[20], lam[64], dl[64], ca[64], out[20 * 4];
    unsigned char sample[64] = {0};
    for (int i = 0; i < 64; ++i) { lam[i] = 1.0 - 0.001 * i; dl[i] = -1e-3; ca[i] = 0.1 + 0.01 * i; sample[i] = i % 16 == 15; }
    orc_rdq20mf_rhs(p, u, 1.0, 0.0, 0.5, du);
    orc_rdq20mf_trajectory(p, u, 64, 1e-2, lam, dl, ca, sample, out);
    for (double vel : {0.0, 2e-3}) {
        double Q[20], Qk[20], a[20], b[20], Q2[20], a2[20], b2[20];
        for (int k = 0; k < 20; ++k) Q[k] = Qk[k] = Q2[k] = u[k];
        int it = 0; double rn = 0;
        if (orc_rdq20mf_local_solve_rate(p, Q, Qk, 0.98, vel, 0.6, 0.5, 1e-12, 30, a, b, &it, &rn)) return 20;
        double qh[20], qkh[20];
        for (int k = 0; k < 20; ++k) { qh[k] = Q2[k]; qkh[k] = Qk[k]; }
        if (tbk::rdq20_local_solve_host(P, qh, qkh, 0.98, vel, 0.6, 0.5, 1e-12, 30, a2, &it, &rn, b2)) return 21;
        for (int k = 0; k < 20; ++k) if (std::fabs(qh[k] - Q[k]) > 1e-12 || std::fabs(a2[k] - a[k]) > 1e-9 || std::fabs(b2[k] - b[k]) > 1e-9) return 22;
    }
    // energies: Hill framework + prestress through the hyper-dual pair evaluation, all 45 pairs
    tbk::EnergyParams e{};
    e.energy = 0; e.penalty = 0;
    const double ho[9] = {0.059, 8.023, 18.472, 16.026, 2.481, 11.120, 0.216, 11.436, 0.0};
    for (int i = 0; i < 9; ++i) e.p[i] = ho[i];
    e.u[0] = 4.0; e.Ta = 0.7; e.hill = tbk::HILL_EXTENDED; e.act_energy = 7; e.act_penalty = 1; e.ap[0] = 10.0; e.adg = tbk::ADG_RLRSQ; e.kappa = 0.5;
    e.sarc = tbk::SARC_PELCE_SUN_LANGEVELD; e.sp[0] = 3.0; e.sp[1] = 0.7;
    e.prestressed = 1;
    const double G[9] = {1.1, 0.2, -0.1, 0.1, 0.9, 0.0, 0.0, 0.1, 1.0};
    for (int i = 0; i < 9; ++i) e.G[i] = G[i];
    const double F[9] = {1.05, 0.02, -0.01, 0.03, 0.97, 0.04, -0.02, 0.01, 1.02}, f0[3] = {1, 0, 0}, s0[3] = {0, 1, 0}, n0[3] = {0, 0, 1};
    double acc = 0.0;
    for (int pr = 0; pr < 45; ++pr) { int mm, nn; tbk::pair_components(pr, mm, nn); acc += tbk::energy_pair(e, F, mm, nn, f0, s0, n0).ab; }
    if (!(acc == acc)) return 23;
    // oracle: condensed assembly (rate-coupled) on a tiny mesh, Hill framework, prestress
    return 0;
}

int main()
{
    if (int rc = sarcomere_and_materials()) { fprintf(stderr, "sarcomere / material section failed: %d\n", rc); return rc; }
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
                for (int model : {0, 1, 2, 3, 5}) { // FHN, Aliev–Panfilov, PCG2019, TT06, O'Hara–Rudy (4 reads point coordinates: covered by the parity tests)
                    double p[64], u0[48];
                    orc_cell_default_params(model, p);
                    orc_cell_default_state(model, p, u0);
                    const int ns = orc_cell_nstates(model);
                    std::vector<double> u((size_t)ns * 37), du(u.size());
                    for (int s = 0; s < ns; ++s) for (int k = 0; k < 37; ++k) u[s * 37 + k] = u0[s];
                    for (int th : {1, 3}) if (orc_reaction_step(model, p, u.data(), du.data(), 37, 0, 0.0, 1e-3, 4, 0.05, th)) return 13;
                    if (model == 2 || model == 3 || model == 5) // the gated models also through the Rush–Larsen step
                        for (int th : {1, 3}) if (orc_reaction_step_rl(model, p, u.data(), 37, 0, 0.0, 1e-3, th)) return 17;
                }
            }
            if (ncomp == 3) {
                std::vector<double> u(nd), nz(nnz), r(nd);
                for (int64_t i = 0; i < nd; ++i) u[i] = 1e-2 * ((i * 7919) % 13 - 6) / 6.0;
                const double p[9] = {0.059, 8.023, 18.472, 16.026, 2.581, 11.120, 0.216, 11.436, 1.0}, fsn[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
                if (orc_assemble_hyperelastic(&om, p, fsn, u.data(), rp.data(), ci.data(), nz.data(), r.data(), 1, nullptr, 0)) return 14;
                { // condensed internal variable, rate-coupled, then Hill framework and prestress
                    const double sp[17] = {1.25, 1.65, 0.18, 2.2, 2.0, 0.381, -0.571, 10.0, 12.0, 0.1, 0.013, 0.13431, 25.184, 0.032653, 0.000778, 22.894e3, 1.0e-6};
                    const int nq = kind == TB_HEX27 ? 27 : 8;
                    const int64_t npts = (int64_t)cd.size() / (nb * ncomp) * nq;
                    std::vector<double> Q(20 * npts, 0.0), Qk(20 * npts, 0.0), up(nd, 0.0);
                    for (int64_t i = 0; i < npts; ++i) Q[i] = Qk[i] = 1.0;
                    std::vector<int> status(npts);
                    orc_set_active_tension(0.6, nullptr);
                    orc_set_condensation(sp, 50.0, Q.data(), Qk.data(), npts, 0.5, 1e-10, 20, status.data());
                    orc_set_condensation_rate(up.data());
                    if (orc_assemble_hyperelastic(&om, p, fsn, u.data(), rp.data(), ci.data(), nz.data(), r.data(), 2, nullptr, 0)) return 16;
                    orc_set_condensation(nullptr, 0, nullptr, nullptr, 0, 1, 0, 1, nullptr);
                    const double ap[12] = {10.0};
                    const double sarc[2] = {3.0, 0.7};
                    orc_set_hill(2, 7, 1, ap, 0, 0.0, 0, sarc);
                    const double G[9] = {1.1, 0.2, -0.1, 0.1, 0.9, 0.0, 0.0, 0.1, 1.0};
                    orc_set_prestress(G);
                    if (orc_assemble_hyperelastic(&om, p, fsn, u.data(), rp.data(), ci.data(), nz.data(), r.data(), 1, nullptr, 0)) return 17;
                    orc_set_prestress(nullptr);
                    orc_set_hill(0, 0, 0, ap, 0, 0.0, 0, sarc);
                    orc_set_active_tension(0.0, nullptr);
                }
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
