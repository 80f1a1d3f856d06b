"""Pins the CPU oracle: (1) against every known-answer vector the reference's own tests hold for the path
(tests/golden/coefficients.json ← Thunderbolt.jl test/test_coefficients.jl:24-188), (2) against the
closed-form identities of SURVEY §8(c), (3) against the closed forms the reference tests assert for the
steppers (test/test_time_integrator.jl:280-296) and layouts (test/test_solution_variables.jl:113-127)."""
import json
import os

import numpy as np
import pytest

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "coefficients.json")))


def line_N(o, xi):
    return o.shape(o.LINE2, [xi])[0]


def test_golden_file_is_reproducible(tmp_path):
    import subprocess
    import sys
    here = os.path.join(os.path.dirname(__file__), "golden")
    before = open(os.path.join(here, "coefficients.json")).read()
    subprocess.check_call([sys.executable, os.path.join(here, "make_coefficients_golden.py")])
    assert open(os.path.join(here, "coefficients.json")).read() == before


@pytest.mark.parametrize("case", [c for c in GOLD["cases"] if "expect" in c and c["name"] in
                                  ("field_scalar", "field_vector", "cartesian", "analytical_norm_plus_t")],
                         ids=lambda c: c["name"])
def test_interpolated_coefficients(oracle, case):
    o = oracle
    cells, xi = GOLD["mesh"]["cells"], GOLD["mesh"]["xi"]
    for e in case["expect"]:
        N = line_N(o, xi[e["qp"]])
        if case["name"].startswith("field"):
            got = o.eval_field(N, np.asarray(case["data"][e["cell"]], dtype=float).reshape(2, -1))
        else:
            x = o.eval_cartesian(N, cells[e["cell"]])
            got = x if case["name"] == "cartesian" else np.array([np.linalg.norm(x) + e["t"]])
        np.testing.assert_allclose(got, np.atleast_1d(e["out"]), rtol=1e-14, atol=1e-15)


def test_spectral_and_diffusivity(oracle):
    o = oracle
    for c in GOLD["cases"]:
        if c["name"] == "spectral_transverse":
            np.testing.assert_allclose(o.eval_spectral([c["f"]], c["lambda"]), c["out"], atol=1e-15)
        elif c["name"] == "spectral_planar":
            np.testing.assert_allclose(o.eval_spectral([c["f"], c["s"]], c["lambda"]), c["out"], atol=1e-15)
        elif c["name"] == "conductivity_to_diffusivity":
            k = o.eval_spectral([c["f"]], c["lambda"])
            np.testing.assert_allclose(o.conductivity_to_diffusivity(k, c["Cm"], c["chi"]), c["out"], atol=1e-15)
        elif c["name"] == "homogeneous_data":
            for e in c["expect"]:
                assert c["data"][o.homogeneous_data_index(c["timings"], e["t"])] == e["out"]


@pytest.mark.parametrize("kind,key,order", [("HEX8", "hex8", 2), ("TET4", "tet4", 2)])
def test_distorted_cell_identities(oracle, kind, key, order):
    """Cells of test_coefficients.jl:195-218: partition of unity, Σ∇N = 0, Σ x⊗∇N = I, Σ detJ·w = volume."""
    o = oracle
    k = getattr(o, kind)
    X = np.array(GOLD["distorted_cells"][key])
    xi, w = o.quadrature(k, order)
    vol = 0.0
    for q in range(len(w)):
        N, dN = o.shape(k, xi[q])
        rc, J, det, Jinv = o.mapping(X, dN)
        assert rc == 0 and det > 0
        dNdx = dN @ Jinv
        np.testing.assert_allclose(N.sum(), 1.0, atol=1e-15)
        np.testing.assert_allclose(dNdx.sum(0), 0.0, atol=1e-14)
        np.testing.assert_allclose(X.T @ dNdx, np.eye(3), atol=1e-13)
        np.testing.assert_allclose(J @ Jinv, np.eye(3), atol=1e-13)
        vol += det * w[q]
    if kind == "TET4":
        ref = abs(np.linalg.det(X[1:] - X[0])) / 6.0
        np.testing.assert_allclose(vol, ref, rtol=1e-14)
    else:  # divergence theorem with a finer rule: the 2-point rule is exact for trilinear-map volumes
        xi4, w4 = o.quadrature(k, 4)
        ref = sum(o.mapping(X, o.shape(k, xi4[q])[1])[2] * w4[q] for q in range(len(w4)))
        np.testing.assert_allclose(vol, ref, rtol=1e-13)


def test_affine_hex_closed_forms(oracle):
    """Mₑ = ρh³/216·[8,4,2,4,4,2,1,2;…], Kₑ(D=I) = −(h/12)·[4,0,−1,0,0,−1,−1,−1;…]."""
    o = oracle
    h = 0.37
    xyz, conn = o.generate_grid_hex(1, 1, 1, (0, 0, 0), (h, h, h))
    cd, nd = o.close_dofs(o.HEX8, 1, conn, len(xyz))
    m = o.Mesh(o.HEX8, 2, xyz, conn, cd)
    rho = 2.5
    Me = o.element_matrix(m, 0, o.Coef(o.COEF_CONST_SCALAR, [rho]), 0)
    np.testing.assert_allclose(Me[0], rho * h ** 3 / 216 * np.array([8, 4, 2, 4, 4, 2, 1, 2.0]), rtol=1e-13)
    np.testing.assert_allclose(Me.sum(), rho * h ** 3, rtol=1e-13)
    np.testing.assert_allclose(Me, Me.T, atol=1e-18)
    Ke = o.element_matrix(m, 1, o.Coef(o.COEF_CONST_TENSOR, np.eye(3).ravel()), 0)
    np.testing.assert_allclose(Ke[0], -(h / 12) * np.array([4, 0, -1, 0, 0, -1, -1, -1.0]), atol=1e-15)
    np.testing.assert_allclose(Ke.sum(1), 0, atol=1e-15)
    # linear form with f ≡ 1 equals Mₑ·1 (§8(c)(3))
    be = o.element_source(m, 0, o.SRC_CONST, [1.0])
    np.testing.assert_allclose(be, o.element_matrix(m, 0, o.Coef(o.COEF_CONST_SCALAR, [1.0]), 0).sum(1), rtol=1e-14)
    # composite cache = 2× (test/test_elements.jl:81-97): linearity in the coefficient
    np.testing.assert_allclose(o.element_matrix(m, 0, o.Coef(o.COEF_CONST_SCALAR, [2 * rho]), 0), 2 * Me, rtol=1e-15)


def test_benchmark_linear_form_vs_high_order(oracle):
    """benchmarks/benchmarks-linear-form.jl: f = ‖x‖+t on generate_grid(Hexahedron,(1,1,1)); the 2-point rule is
    compared with a 4-point rule on the same smooth integrand away from the origin."""
    o = oracle
    xyz, conn = o.generate_grid_hex(1, 1, 1, (1, 1, 1), (2, 2, 2))
    cd, _ = o.close_dofs(o.HEX8, 1, conn, len(xyz))
    b2 = o.element_source(o.Mesh(o.HEX8, 2, xyz, conn, cd), 0, o.SRC_NORM_PLUS_T, t=0.5)
    b4 = o.element_source(o.Mesh(o.HEX8, 4, xyz, conn, cd), 0, o.SRC_NORM_PLUS_T, t=0.5)
    np.testing.assert_allclose(b2, b4, rtol=1e-3)
    np.testing.assert_allclose(b2.sum(), b4.sum(), rtol=1e-4)


def test_spectral_field_matches_constant_frame(oracle):
    o = oracle
    xyz, conn = o.generate_grid_hex(2, 1, 1, (0, 0, 0), (1, 1, 1))
    cd, _ = o.close_dofs(o.HEX8, 1, conn, len(xyz))
    m = o.Mesh(o.HEX8, 2, xyz, conn, cd)
    f, s, n = np.array([1, 1, 0.0]), np.array([-1, 1, 0.0]), np.array([0, 0, 3.0])
    lam = [3.0, 2.0, 1.0]
    fld = np.tile(np.stack([2 * f, 0.5 * s, n]), (2, 8, 1, 1))  # unnormalised, orthogonal → same frame after normalising
    Kf = o.element_matrix(m, 1, o.Coef(o.COEF_SPECTRAL_FIELD, lam, field=fld, Cm=2.0, chi=0.5, wrap=True), 1)
    fn, sn, nn = (v / np.linalg.norm(v) for v in (f, s, n))
    Kc = o.element_matrix(m, 1, o.Coef(o.COEF_SPECTRAL_CONST, np.concatenate([fn, sn, nn, lam])), 1)
    np.testing.assert_allclose(Kf, Kc, rtol=1e-12, atol=1e-15)


def test_orthogonalize(oracle):
    f, s, n = oracle.orthogonalize([2, 0, 0.0], [1, 1, 0.0], [1, 1, 1.0])
    np.testing.assert_allclose(f, [1, 0, 0], atol=1e-15)
    assert abs(f @ s) < 1e-15 and abs(f @ n) < 1e-15
    # src/utils.jl:131-139 does NOT renormalise w₂ before projecting v₃ on it — transcribed as written:
    v2, v3 = np.array([1, 1, 0.0]) / np.sqrt(2), np.array([1, 1, 1.0]) / np.sqrt(3)
    w2 = v2 - (f @ v2) * f
    np.testing.assert_allclose(s, w2, atol=1e-15)
    np.testing.assert_allclose(n, v3 - (f @ v3) * f - (w2 @ v3) * w2, atol=1e-15)


def test_cell_models_closed_form(oracle):
    o = oracle
    # FHN cells/fhn.jl:21-34
    p = o.cell_default_params(o.CELL_FHN)
    np.testing.assert_allclose(p, [0.1, 0.5, 1.0, 0.0, 0.01, 1.0])
    phi, s = 0.7, 0.2
    np.testing.assert_allclose(o.cell_rhs(o.CELL_FHN, p, [phi, s]),
                               [phi * (1 - phi) * (phi - 0.1) - s, 0.01 * (0.5 * phi - s)], rtol=1e-15)
    # Aliev–Panfilov cells/aliev-panfilov.jl:17-31, state order (s, φₘ)
    p = o.cell_default_params(o.CELL_ALIEV_PANFILOV)
    ct, k, a, e0, m1, m2 = p
    eps = e0 + s * m1 / (phi + m2)
    np.testing.assert_allclose(o.cell_rhs(o.CELL_ALIEV_PANFILOV, p, [s, phi]),
                               [ct * eps * (-s - k * phi * (phi - a - 1.0)), ct * (k * phi * (phi - 1.0) * (phi - a) - phi * s)],
                               rtol=1e-15)
    # PCG2019: default initial state is a fixed point of every gate (pcg2019.jl:137-152)
    p = o.cell_default_params(o.CELL_PCG2019)
    u0 = o.cell_default_state(o.CELL_PCG2019, p)
    assert u0[0] == -85.0
    du = o.cell_rhs(o.CELL_PCG2019, p, u0)
    np.testing.assert_allclose(du[1:], 0.0, atol=1e-18)
    sig = lambda phi, E, k, sg: 1 / (1 + np.exp(sg * (phi - E) / k))  # noqa: E731
    np.testing.assert_allclose([u0[1], u0[2]], [sig(-85.0, -78.7, 5.93, 1.0), sig(-85.0, -52.244, 6.5472, -1.0)], rtol=1e-15)
    # I_Na = g_Na m³ h² (φ−E_Na), pcg2019.jl:77
    u = u0.copy(); u[0] = -20.0
    du = o.cell_rhs(o.CELL_PCG2019, p, u)
    phi = -20.0
    I = (12.0 * u[2] ** 3 * u[1] ** 2 * (phi - 65.0) + 0.73893 * sig(phi, -91.9655, 12.4997, 1) * (phi + 85)
         + 0.1688 * sig(phi, 14.3116, 11.462, -1) * u[4] * (phi + 85) + 0.11503 * sig(phi, 0.7, 4.3, -1) * u[3] * (phi - 50)
         + 0.056 * u[6] * sig(phi, -49.6, 23.5, 1) * (phi + 85) + 0.008 * u[5] * (phi + 85))
    np.testing.assert_allclose(du[0], -I, rtol=1e-14)
    tau_h = 2 * 6.80738 * np.exp(0.799163 * (phi + 78.7) / 5.93) / (1 + np.exp((phi + 78.7) / 5.93))
    np.testing.assert_allclose(du[1], (sig(phi, -78.7, 5.93, 1) - u[1]) / tau_h, rtol=1e-14)


def test_forward_euler_and_layouts(oracle):
    o = oracle
    n = 5
    p = o.cell_default_params(o.CELL_PCG2019)
    rng = np.random.default_rng(0)
    pts = np.tile(o.cell_default_state(o.CELL_PCG2019, p), (n, 1)) + rng.uniform(-1e-2, 1e-2, (n, 7))
    soa = np.ascontiguousarray(pts.T).ravel().copy()   # u[k + s·npoints]  (solution_variables.jl:60-63)
    aos = pts.ravel().copy()                            # u[k·nstates + s]
    du_s = o.reaction_step(o.CELL_PCG2019, p, soa, n, o.LAYOUT_SOA, dt=0.01)
    du_a = o.reaction_step(o.CELL_PCG2019, p, aos, n, o.LAYOUT_AOS, dt=0.01)
    np.testing.assert_array_equal(soa.reshape(7, n).T, aos.reshape(n, 7))
    np.testing.assert_array_equal(du_s.reshape(7, n).T, du_a.reshape(n, 7))
    for k in range(n):
        np.testing.assert_allclose(aos.reshape(n, 7)[k], pts[k] + 0.01 * o.cell_rhs(o.CELL_PCG2019, p, pts[k]), rtol=1e-15)
    # layout probe of test/test_solution_variables.jl:113-127: uₙmat[k,j] ↔ u[(j-1)·npoints + k]
    u = np.zeros(7 * n)
    u.reshape(7, n)[3, 2] = 1.0
    assert u[3 * n + 2] == 1.0


def test_adaptive_substepper(oracle):
    """partitioned_solver.jl:196-234: below the threshold one full step; above it `substeps` Euler steps."""
    o = oracle
    p = o.cell_default_params(o.CELL_FHN)
    for u0, thr in (([0.5, 0.0], 1e9), ([0.5, 0.0], 1e-9)):
        u = np.array(u0)
        o.reaction_step(o.CELL_FHN, p, u, 1, dt=0.4, substeps=4, threshold=thr)
        ref = np.array(u0)
        if thr > 1:
            ref = ref + 0.4 * o.cell_rhs(o.CELL_FHN, p, ref)
        else:
            for _ in range(4):
                ref = ref + 0.1 * o.cell_rhs(o.CELL_FHN, p, ref)
        np.testing.assert_allclose(u, ref, rtol=1e-15)


def test_heat_algebra(oracle):
    o = oracle
    xyz, conn = o.generate_grid_hex(3, 2, 2, (0, 0, 0), (1, 1, 1))
    cd, nd = o.close_dofs(o.HEX8, 1, conn, len(xyz))
    rp, ci = o.build_pattern(cd, nd)
    m = o.Mesh(o.HEX8, 2, xyz, conn, cd)
    M = o.assemble_matrix(m, 0, o.Coef(o.COEF_CONST_SCALAR, [1.0]), rp, ci)
    K = o.assemble_matrix(m, 1, o.Coef(o.COEF_CONST_TENSOR, np.diag([4.5e-5, 2e-5, 2e-5]).ravel()), rp, ci)
    A = o.heat_matrix(M, K, 0.5)
    np.testing.assert_array_equal(A, M - 0.5 * K)
    one = np.ones(nd)
    # pure-Neumann diffusion keeps u ≡ 1 (test/test_time_integrator.jl:29-41): K·1 = 0, so A·1 = M·1
    np.testing.assert_allclose(o.spmv_csr(rp, ci, K, one), 0, atol=1e-18)
    np.testing.assert_allclose(o.spmv_csr(rp, ci, A, one), o.spmv_csr(rp, ci, M, one), rtol=1e-14)
    np.testing.assert_allclose(o.spmv_csr(rp, ci, M, one).sum(), 1.0, rtol=1e-13)  # ∫1 = volume
    # source f≡1 equals M·1; threaded colour / EA variants agree with the sequential loop
    np.testing.assert_allclose(o.assemble_source(m, o.SRC_CONST, [1.0]), o.spmv_csr(rp, ci, M, one), rtol=1e-13)
    col, nc = o.color_cells(cd, nd)
    for c in range(nc):  # valid colouring: no two cells of a colour share a dof
        d = cd[col == c].ravel()
        assert len(np.unique(d)) == len(d)
    np.testing.assert_allclose(o.assemble_matrix(m, 1, o.Coef(o.COEF_CONST_SCALAR, [1.0]), rp, ci, nthreads=4, color=col, ncolors=nc),
                               o.assemble_matrix(m, 1, o.Coef(o.COEF_CONST_SCALAR, [1.0]), rp, ci), rtol=1e-13, atol=1e-16)
    np.testing.assert_allclose(o.assemble_source(m, o.SRC_COS_EXP, t=0.1, nthreads=4), o.assemble_source(m, o.SRC_COS_EXP, t=0.1), rtol=1e-13)


def test_q2_tables(oracle):
    o = oracle
    xi = [0.3, -0.2, 0.55]
    N, dN = o.shape(o.HEX27, xi)
    np.testing.assert_allclose(N.sum(), 1.0, atol=1e-14)
    np.testing.assert_allclose(dN.sum(0), 0.0, atol=1e-14)
    # delta property at the Ferrite reference coordinates (vertices, edge midpoints, face centres, centre)
    ref = np.array([[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1],
                    [0, -1, -1], [1, 0, -1], [0, 1, -1], [-1, 0, -1], [0, -1, 1], [1, 0, 1], [0, 1, 1], [-1, 0, 1],
                    [-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0],
                    [0, 0, -1], [0, -1, 0], [1, 0, 0], [0, 1, 0], [-1, 0, 0], [0, 0, 1], [0, 0, 0]], dtype=float)
    for a in range(27):
        np.testing.assert_allclose(o.shape(o.HEX27, ref[a])[0], np.eye(27)[a], atol=1e-14)
    # finite-difference check of the gradients
    e = 1e-6
    for d in range(3):
        xp, xm = np.array(xi), np.array(xi)
        xp[d] += e; xm[d] -= e
        np.testing.assert_allclose((o.shape(o.HEX27, xp)[0] - o.shape(o.HEX27, xm)[0]) / (2 * e), dN[:, d], atol=1e-8)


def test_heterogeneous_fhn_reduces_to_fhn_and_reads_x(oracle):
    """HeterogeneousFHNModel of the reference's how-to (docs/src/literate-howto/custom-ep-cell-model.jl:43-56): with a constant e it is the FHN
    right-hand side (cells/fhn.jl:21-34, f = 1) bit for bit; with a gradient only du[2] changes, by g·x·(bφ − cs − d)."""
    o = oracle
    rng = np.random.default_rng(3)
    n = 17
    u = rng.uniform(-0.2, 1.0, size=2 * n)
    xs = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)
    pf = o.cell_default_params(o.CELL_FHN)
    ph = o.cell_default_params(o.CELL_FHN_HETEROGENEOUS)
    a, b = u.copy(), u.copy()
    da = o.reaction_step(o.CELL_FHN, pf, a, n, o.LAYOUT_SOA, t=0.3, dt=0.05)
    db = o.reaction_step_x(o.CELL_FHN_HETEROGENEOUS, ph, b, n, xs, o.LAYOUT_SOA, t=0.3, dt=0.05)
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(da, db)
    ph[5:8] = [0.02, -0.01, 0.005]
    c = u.copy()
    dc = o.reaction_step_x(o.CELL_FHN_HETEROGENEOUS, ph, c, n, xs, o.LAYOUT_SOA, t=0.3, dt=0.05)
    phi, s = u[:n], u[n:]
    np.testing.assert_array_equal(dc[:n], da[:n])
    np.testing.assert_allclose(dc[n:] - da[n:], (xs.astype(float) @ ph[5:8]) * (0.5 * phi - 1.0 * s - 0.0), rtol=1e-12, atol=1e-16)


def test_pcg2019_rush_larsen_is_exact_for_gates_and_first_order(oracle):
    """Rush–Larsen for the reference's PCG2019 (gates relax as (g∞ − g)/τ_g, cells/pcg2019.jl:96-118): a gate with frozen φₘ follows
    g∞ + (g − g∞)e^{−Δt/τ}; for Δt → 0 the step tends to forward Euler at second order in Δt."""
    o = oracle
    p = o.cell_default_params(o.CELL_PCG2019)
    u0 = o.cell_default_state(o.CELL_PCG2019, p)
    u0 = u0 + np.array([25.0, -0.1, 0.2, -0.1, 0.05, 0.1, 0.2])
    du = o.cell_rhs(o.CELL_PCG2019, p, u0)
    dt = 0.5
    u = u0.copy()
    o.reaction_step_rl(o.CELL_PCG2019, p, u, 1, o.LAYOUT_SOA, dt=dt)
    tau_m = p[3]
    m_inf = u0[2] + du[2] * tau_m
    np.testing.assert_allclose(u[2], m_inf + (u0[2] - m_inf) * np.exp(-dt / tau_m), rtol=1e-13)
    np.testing.assert_allclose(u[0], u0[0] + dt * du[0], rtol=1e-15)        # φₘ by forward Euler
    errs = []
    for h in (1e-2, 5e-3):
        a, b = u0.copy(), u0.copy()
        o.reaction_step_rl(o.CELL_PCG2019, p, a, 1, o.LAYOUT_SOA, dt=h)
        o.reaction_step(o.CELL_PCG2019, p, b, 1, o.LAYOUT_SOA, dt=h, want_du=False)
        errs.append(np.abs(a - b).max())
    assert errs[1] < 0.3 * errs[0]


def test_baseline_form_of_the_per_colour_assembly_matches_the_canonical_loop(oracle):
    """oracle.AssemblyPlan (bench.py's cpu_baseline: scatter positions looked up once, per-colour cell lists, first-touch zero fill, hexahedron instances of
    the element routines, planned element-assembly source) against the literal sequential loop the parity tests use — same element sums, colour order
    instead of cell order: agreement to rounding for M, K and b, with one thread and with several, and identical bits between repeated assemblies."""
    o = oracle
    xyz, conn = o.generate_grid_hex(7, 6, 5, (0, 0, 0), (1.0, 0.9, 1.2))
    xyz = xyz + 0.02 * np.sin(7.0 * xyz[:, [1, 2, 0]])                       # distorted cells
    cd, nd = o.close_dofs(o.HEX8, 1, conn, len(xyz))
    rp, ci = o.build_pattern(cd, nd)
    col, nc = o.color_cells(cd, nd)
    m = o.Mesh(o.HEX8, 2, xyz, conn, cd)
    cM = o.Coef(o.COEF_CONST_SCALAR, [1.3])
    cK = o.Coef(o.COEF_CONST_TENSOR, np.array([[2.0, 0.3, 0.1], [0.3, 1.5, 0.2], [0.1, 0.2, 1.0]]).ravel())
    refM, refK = o.assemble_matrix(m, 0, cM, rp, ci), o.assemble_matrix(m, 1, cK, rp, ci)
    refb = o.assemble_source(m, o.SRC_COS_EXP, t=0.3)
    for th in (1, 3):
        plan = o.AssemblyPlan(m, rp, ci, col, nc, th)
        nzM, nzK, b = plan.new_values(), plan.new_values(), np.empty(nd)
        plan.assemble(0, cM, nzM); plan.assemble(1, cK, nzK); plan.assemble_source(o.SRC_COS_EXP, b, t=0.3)
        assert np.abs(nzM - refM).max() < 1e-14 * np.abs(refM).max()
        assert np.abs(nzK - refK).max() < 1e-14 * np.abs(refK).max()
        np.testing.assert_array_equal(b, refb)                                   # element assembly sums in cell order: the same bits
        again = plan.new_values()
        plan.assemble(1, cK, again)
        np.testing.assert_array_equal(again, nzK)
