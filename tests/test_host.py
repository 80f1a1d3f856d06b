"""CPU-only checks of the shipped library: it loads, exports every symbol include/tbhip.h declares, its
host-side generators reproduce the oracle's Ferrite-convention mesh / dof numbering / sparsity graph
BIT-EXACTLY (integer parity), and argument errors are reported through the C-ABI error channel."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(tb):
    hdr = open(os.path.join(ROOT, "include", "tbhip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(tb_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 40
    lib = C.CDLL(tb._lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), "libtbhip.so does not export %s" % name
    assert declared == set(tb._lib.SIGNATURES), declared ^ set(tb._lib.SIGNATURES)
    assert b"gfx950" in tb.lib().tb_version()


def test_missing_library_fails_loudly(tb, monkeypatch):
    monkeypatch.setattr(tb._lib, "_lib", None)
    monkeypatch.setattr(tb._lib, "LIB_PATH", "/nonexistent/libtbhip.so")
    with pytest.raises(ImportError, match="no CPU fallback"):
        tb._lib.lib()


@pytest.mark.parametrize("nel", [(1, 1, 1), (3, 2, 4), (5, 5, 5)])
def test_grid_dofs_pattern_bit_exact(tb, oracle, nel):
    g = tb.generate_mesh(tb.Hexahedron, nel, (-1, -1, -1), (1, 1, 1))
    xyz, conn = oracle.generate_grid_hex(*nel, (-1, -1, -1), (1, 1, 1))
    np.testing.assert_array_equal(g.conn, conn)
    np.testing.assert_array_equal(g.xyz, xyz)
    # Ferrite generate_grid: first cell of the default box, local vertex order of src/mesh/generators.jl:62-79
    px, py = nel[0] + 1, nel[1] + 1
    assert list(conn[0]) == [0, 1, px + 1, px, px * py, px * py + 1, px * py + px + 1, px * py + px]
    for kind, okind, order, ncomp in ((tb._lib.TB_HEX8, oracle.HEX8, 1, 1), (tb._lib.TB_HEX8, oracle.HEX8, 1, 3),
                                      (tb._lib.TB_HEX27, oracle.HEX27, 2, 1), (tb._lib.TB_HEX27, oracle.HEX27, 2, 3)):
        dh = tb.DofHandler(g, tb.LagrangeCollection(order) ** ncomp)
        cd, nd = oracle.close_dofs(okind, ncomp, conn, len(xyz))
        assert dh.ndofs == nd
        np.testing.assert_array_equal(dh.cell_dofs, cd)
        if order == 1 and ncomp == 1:
            assert nd == len(xyz)
            assert list(cd[0]) == list(range(8))  # first cell numbers its vertices 1..8 (close!: first visit)
        sp = tb.allocate_matrix(dh)
        rp, ci = oracle.build_pattern(cd, nd)
        np.testing.assert_array_equal(sp.rowptr, rp)
        np.testing.assert_array_equal(sp.colidx, ci)
    n = nel
    dh = tb.DofHandler(g)
    assert tb.allocate_matrix(dh).nnz == (3 * n[0] + 1) * (3 * n[1] + 1) * (3 * n[2] + 1)  # SURVEY §8 sizes


def test_q2_dof_count(tb):
    g = tb.generate_mesh(tb.Hexahedron, (4, 3, 2))
    dh = tb.DofHandler(g, tb.LagrangeCollection(2) ** 3)
    assert dh.ndofs == 3 * 9 * 7 * 5
    assert dh.cell_dofs.shape == (24, 81)
    # vector field: node-major, component-minor (src/ferrite-addons/io.jl:233-238)
    assert list(dh.cell_dofs[0][:6]) == [0, 1, 2, 3, 4, 5]


def test_perturbation_keeps_boundary_and_orientation(tb, oracle):
    g = tb.generate_mesh(tb.Hexahedron, (6, 6, 6), (0, 0, 0), (1, 1, 1), perturb=0.2)
    g0 = tb.generate_mesh(tb.Hexahedron, (6, 6, 6), (0, 0, 0), (1, 1, 1))
    on_bnd = np.any((g0.xyz == 0) | (g0.xyz == 1), axis=1)
    assert np.abs(g.xyz[on_bnd] - g0.xyz[on_bnd]).max() < 1e-15
    assert np.abs(g.xyz - g0.xyz).max() > 1e-3
    dh = tb.DofHandler(g)
    m = oracle.Mesh(oracle.HEX8, 2, g.xyz, g.conn, dh.cell_dofs)
    vol = sum(oracle.element_matrix(m, 0, oracle.Coef(oracle.COEF_CONST_SCALAR, [1.0]), c).sum() for c in range(g.n_cells))
    np.testing.assert_allclose(vol, 1.0, rtol=1e-12)


def test_error_channel_without_gpu(tb):
    lib = tb.lib()
    assert lib.tb_host_generate_grid_hex(0, 1, 1, None, None, None, None) == tb._lib.TB_ERR_BAD_ARG
    assert b"bad argument" in lib.tb_last_error_string()
    ns, npar, phi = C.c_int(), C.c_int(), C.c_int()
    assert lib.tb_cell_model_info(99, C.byref(ns), C.byref(npar), C.byref(phi)) == tb._lib.TB_ERR_BAD_ARG
    assert lib.tb_cell_model_info(tb._lib.TB_CELL_ALIEV_PANFILOV, C.byref(ns), C.byref(npar), C.byref(phi)) == 0
    assert (ns.value, npar.value, phi.value) == (2, 6, 1)
    assert lib.tb_reaction_step(None, 0, None, 0, None, None, 0, 0, 0, 0.0, 0.0, 1, 0.0) == tb._lib.TB_ERR_BAD_ARG


def test_cell_model_defaults_match_oracle(tb, oracle):
    for cls, oid in ((tb.FHNModel, oracle.CELL_FHN), (tb.AlievPanfilovModel, oracle.CELL_ALIEV_PANFILOV),
                     (tb.PCG2019, oracle.CELL_PCG2019), (tb.TT06, oracle.CELL_TT06)):
        m = cls()
        np.testing.assert_array_equal(m.params, oracle.cell_default_params(oid))
        np.testing.assert_allclose(m.default_initial_state(), oracle.cell_default_state(oid), rtol=1e-15)
        assert tb.num_states(m) == oracle.cell_nstates(oid)
    assert tb.transmembranepotential_index(tb.AlievPanfilovModel()) == 2  # state order (s, φₘ)
    assert tb.FHNModel(a=0.3).params[0] == 0.3


def test_slab_partition(tb):
    D = tb.distributed
    covered = []
    for r in range(3):
        z0, z1 = D.slab_range(8, 3, r)
        covered += list(range(z0, z1))
    assert covered == list(range(8))
    p = D.SlabPartition((4, 4, 8), (0, 0, 0), (1, 1, 2), 2, 1)
    assert p.local_nel() == (4, 4, 4) and p.left[2] == 1.0 and p.right[2] == 2
    lo, up = p.interface_nodes()
    assert up is None and len(lo) == 25


def test_reaction_tangent_controller_stepsize_known_answers(tb):
    """σ(R) of the ReactionTangentController at the values the reference pins (test/test_os_gearing.jl:250-296):
    R = 0.5 with σ_s = 0.5, σ_c = 1 on Δt ∈ (0.05π, 0.2π), and the σ_s = ∞ step function incl. the boundary R = σ_c."""
    dt = 0.1 * np.pi
    bounds = (dt * 0.5, dt * 2)
    rtc = tb.ReactionTangentController(None, 0.5, 1.0, bounds)
    expected = (1 - 1 / (1 + np.exp((1.0 - 0.5) * 0.5))) * (bounds[1] - bounds[0]) + bounds[0]
    assert rtc.stepsize(0.5) == pytest.approx(expected, rel=1e-15)
    assert bounds[0] < rtc.stepsize(0.5) < bounds[1]
    for sigma_c, want in ((0.75, bounds[1]), (0.5, bounds[1]), (0.25, bounds[0])):
        assert tb.ReactionTangentController(None, np.inf, sigma_c, bounds).stepsize(0.5) == want
    # monotone: faster reactions → shorter steps
    rs = np.linspace(-2, 5, 50)
    dts = [rtc.stepsize(r) for r in rs]
    assert all(a >= b for a, b in zip(dts, dts[1:]))


def test_quadrilateral_grid_conventions(tb, oracle):
    """generate_grid(Quadrilateral, (nx, ny), left, right): nodes x-fastest on the lattice, counter-clockwise cells, z = 0;
    first-visit dof numbering and the sparsity graph equal the oracle's on the same connectivity."""
    g = tb.generate_mesh(tb.Quadrilateral, (3, 2), (-1.0, -1.0), (1.0, 1.0))
    assert g.xyz.shape == (12, 3) and g.conn.shape == (6, 4) and np.all(g.xyz[:, 2] == 0)
    np.testing.assert_array_equal(g.conn[0], [0, 1, 5, 4])
    np.testing.assert_allclose(g.xyz[5], [-1 + 2 / 3, 0.0, 0.0])
    x = g.xyz[g.conn]                                           # positive orientation everywhere
    a, b = x[:, 1, :2] - x[:, 0, :2], x[:, 3, :2] - x[:, 0, :2]
    area2 = a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]
    assert np.all(area2 > 0)
    dh = tb.DofHandler(g)
    cd, nd = oracle.close_dofs(oracle.QUAD4, 1, g.conn, g.n_nodes)
    np.testing.assert_array_equal(dh.cell_dofs, cd)
    assert dh.ndofs == nd == 12
    sp = tb.allocate_matrix(dh)
    rp, ci = oracle.build_pattern(cd, nd)
    np.testing.assert_array_equal(sp.rowptr, rp)
    np.testing.assert_array_equal(sp.colidx, ci)
    # the oracle's 2-D element on this mesh: Σ M = area, K·1 = 0
    om = oracle.Mesh(oracle.QUAD4, 2, np.ascontiguousarray(g.xyz[:, :2]), g.conn, cd)
    M = oracle.assemble_matrix(om, 0, oracle.Coef(oracle.COEF_CONST_SCALAR, [1.0]), rp, ci)
    assert M.sum() == pytest.approx(4.0, rel=1e-14)
    K = oracle.assemble_matrix(om, 1, oracle.Coef(oracle.COEF_CONST_TENSOR, [2.0, 0.3, 0.3, 1.0]), rp, ci)
    assert np.abs(oracle.spmv_csr(rp, ci, K, np.ones(nd))).max() < 1e-14


def test_deuflhard_continuation_controllers(tb):
    """test/test_time_integrator.jl:315-398: the three load-path step controllers are pure functions of Newton's contraction history, so
    prescribed Θₖ pin their laws exactly — acceptance, the shrink on rejection, and the predictor with its three denominators
    (2Θ₀, g(Θ₀), the mean); g(x) = √(1 + 4x) − 1."""
    g = lambda x: np.sqrt(1 + 4 * x) - 1                                          # noqa: E731
    ctrls = (tb.Deuflhard2004DiscreteContinuationController(theta_min=1 / 8, p=1),
             tb.Deuflhard2004_B_DiscreteContinuationControllerVariant(theta_min=1 / 8, p=1),
             tb.ExperimentalDiscreteContinuationController(theta_min=1 / 8, p=1))
    for c in ctrls:
        assert c.should_accept_step([0.1, 0.2])
        assert not c.should_accept_step([0.1, c.theta_reject + 0.01])
        assert c.should_accept_step([10.0], enforce_monotonic_convergence=False)     # without monotonicity only finiteness matters
        assert not c.should_accept_step([0.1, np.nan], enforce_monotonic_convergence=False)
    c = ctrls[0]
    clamp = lambda q, c: min(max(q, c.qmin), c.qmax)                               # noqa: E731
    dt = c.reject_step(0.4, [0.1, 2.0])
    assert dt == pytest.approx(clamp(c.gamma * (g(c.theta_bar) / g(2.0)) ** (1 / c.p), c) * 0.4) and dt < 0.4
    assert c.reject_step(0.4, [0.1, 0.2]) == 0.4                                    # nothing above the threshold: dt untouched
    th = [0.3, 0.4]
    t0 = max(th[0], c.theta_min)
    assert ctrls[0].adapt_dt(0.4, th) == pytest.approx(clamp(c.gamma * (g(c.theta_bar) / (2 * t0)) ** (1 / c.p), c) * 0.4)
    assert ctrls[1].adapt_dt(0.4, th) == pytest.approx(clamp(c.gamma * (g(c.theta_bar) / g(t0)) ** (1 / c.p), c) * 0.4)
    e = ctrls[2]
    t0e = max(sum(th) / len(th), e.theta_min)
    assert e.adapt_dt(0.4, th) == pytest.approx(clamp(e.gamma * (g(e.theta_bar) / (2 * t0e)) ** (1 / e.p), e) * 0.4)
    assert ctrls[0].adapt_dt(0.4, []) == pytest.approx(clamp(c.gamma * (g(c.theta_bar) / (2 * c.theta_min)) ** (1 / c.p), c) * 0.4)   # empty history → Θmin
    assert (e.theta_reject, e.theta_bar) == (0.9, 0.75) and (c.theta_reject, c.theta_bar, c.gamma, c.qmin, c.qmax) == (0.95, 0.5, 0.95, 0.2, 5.0)


def test_eisenstat_walker_forcing_law(tb):
    """newton_raphson.jl:158-178: η₀ on the first step, then γ (‖rₖ‖/‖rₖ₋₁‖)^α with the safeguard γ ηₖ₋₁^α when that exceeds the threshold,
    clamped to [0, ηmax]."""
    f = tb.EisenstatWalkerForcing()
    assert f.prestep(1.0, 0) == 0.5
    assert f.prestep(0.5, 1) == pytest.approx(max(0.9 * 0.25, 0.9 * 0.5 ** 2))      # both 0.225: safeguard 0.225 > 0.1 and not larger
    eta2 = f.prestep(0.05, 2)                                                       # fast drop: γ·0.01 = 0.009, safeguard γ·0.225² = 0.0456 < 0.1 → not applied
    assert eta2 == pytest.approx(0.009)
    f = tb.EisenstatWalkerForcing()
    f.prestep(1.0, 0)
    assert f.prestep(2.0, 1) == 0.9                                                 # growth of the residual saturates at ηmax
    f = tb.EisenstatWalkerForcing(safeguard=False)
    f.prestep(1.0, 0)
    assert f.prestep(0.1, 1) == pytest.approx(0.009)


def test_bench_refuses_a_world_size_that_is_not_gpus():
    """bench.py under a launcher whose WORLD_SIZE differs from --gpus must exit non-zero without printing a result line (the
    argument used to be ignored and a driver scaling run would have recorded n_gpus: 1 eight times)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode != 0 and "n_gpus" not in r.stdout and "WORLD_SIZE" in r.stderr
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "64"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode != 0 and "n_gpus" not in r.stdout           # more ranks than visible GPUs: refused before anything is launched


@pytest.mark.parametrize("nel,order", [((6, 5, 7), 1), ((4, 3, 5), 2)])
def test_locality_permutation_returns_a_shuffled_lattice_to_its_own_numbering(tb, nel, order):
    """tb_host_locality_permutation on a perturbed box whose cells and nodes were renumbered at random: the cells come back in generate_grid's
    order, the nodes in its lexicographic order, and the dofs get the numbers close!(dh) gives on the original grid (reference behaviour to match:
    the cell loop of src/modeling/core/coordinate_systems.jl:145-171 is insensitive to numbering — the permutation only changes names)."""
    g0 = tb.generate_mesh(tb.Hexahedron, nel, (-1, -1, -1), (1, 2, 3), perturb=0.2)
    ip = tb.LagrangeCollection(order)
    dh0 = tb.DofHandler(g0, ip)
    rng = np.random.default_rng(7)
    pn, pc = rng.permutation(g0.n_nodes), rng.permutation(g0.n_cells)
    inv = np.empty_like(pn); inv[pn] = np.arange(g0.n_nodes)
    gs = tb.Grid(tb.Hexahedron, g0.xyz[pn], inv[g0.conn[pc]].astype(np.int32))
    dhs = tb.DofHandler(gs, ip)
    cell_perm, node_perm, dof_perm = tb.locality_permutation(gs, dhs)
    assert sorted(cell_perm) == list(range(gs.n_cells)) and sorted(node_perm) == list(range(gs.n_nodes)) and sorted(dof_perm) == list(range(dhs.ndofs))
    gr = tb.renumber_grid(gs, cell_perm, node_perm)
    np.testing.assert_array_equal(gr.conn, g0.conn)
    np.testing.assert_array_equal(gr.xyz, g0.xyz)
    # renumber!(dh, perm) on the shuffled grid, cells read in the new order = the original dof table
    np.testing.assert_array_equal(dof_perm[dhs.cell_dofs][cell_perm], dh0.cell_dofs)
    # … and a DofHandler closed on the renumbered grid numbers the same way (first visit)
    np.testing.assert_array_equal(tb.DofHandler(gr, ip).cell_dofs, dh0.cell_dofs)


def test_locality_permutation_on_tetrahedra_and_bad_arguments(tb):
    g = tb.generate_mesh(tb.Hexahedron, (3, 3, 3), (0, 0, 0), (1, 1, 1))
    tets = np.concatenate([g.conn[:, [0, 1, 3, 4]], g.conn[:, [1, 2, 3, 6]], g.conn[:, [1, 5, 4, 6]], g.conn[:, [3, 7, 6, 4]], g.conn[:, [1, 3, 4, 6]]]).astype(np.int32)
    gt = tb.Grid(tb.Tetrahedron, g.xyz, tets)
    dh = tb.DofHandler(gt)
    cp, npm, dp = tb.locality_permutation(gt, dh)
    assert sorted(cp) == list(range(gt.n_cells)) and sorted(npm) == list(range(gt.n_nodes)) and sorted(dp) == list(range(dh.ndofs))
    # first visit: the first cell of the new order holds dofs 0 … 3
    assert sorted(dp[dh.cell_dofs[cp[0]]]) == [0, 1, 2, 3]
    L = tb._lib
    rc = tb.lib().tb_host_locality_permutation(99, gt.n_nodes, gt.xyz.ctypes.data_as(L.c_dp), gt.n_cells, gt.conn.ctypes.data_as(L.c_i32p), 0, None, 0, 0,
                                               cp.ctypes.data_as(L.c_i32p), None, None)
    assert rc == L.TB_ERR_BAD_ARG and b"geom_kind" in tb.lib().tb_last_error_string()
