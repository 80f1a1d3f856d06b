"""Mesh readers against the reference's own test data (test/data/{openCARP,mfem,voom2}, committed as
fixtures under tests/golden/meshes) and the assertions of test/test_mesh.jl:96-136: cell types, counts, detJ > 0."""
import os

import numpy as np
import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "meshes")

CASES = [("ref-segment", "Line"), ("ref-triangle", "Triangle"), ("ref-square", "Quadrilateral"), ("ref-tetrahedron", "Tetrahedron"),
         ("ref-cube", "Hexahedron"), ("ref-prism", "Wedge")]


def signed_measure(kind, x):
    """Orientation measure at the first vertex (all reference files hold reference-shaped cells): > 0 ⇔ detJ > 0."""
    if kind == "Line":
        return np.linalg.norm(x[1] - x[0])
    if kind in ("Triangle", "Quadrilateral"):
        a, b = x[1] - x[0], x[-1] - x[0]
        return a[0] * b[1] - a[1] * b[0]
    if kind == "Tetrahedron":
        return np.linalg.det(np.array([x[1] - x[0], x[2] - x[0], x[3] - x[0]]))
    if kind == "Hexahedron":
        return np.linalg.det(np.array([x[1] - x[0], x[3] - x[0], x[4] - x[0]]))
    if kind == "Wedge":
        return np.linalg.det(np.array([x[1] - x[0], x[2] - x[0], x[3] - x[0]]))
    if kind == "Pyramid":
        return np.linalg.det(np.array([x[1] - x[0], x[2] - x[0], x[4] - x[0]]))
    raise AssertionError(kind)


def check_orientation(mg):
    for t, c in zip(mg.cell_types, mg.cells):
        assert signed_measure(t, mg.nodes[list(c)]) > 0, (t, c)


@pytest.mark.parametrize("name,kind", CASES)
def test_opencarp_reference_files(tb, name, kind):
    mg = tb.meshio.load_carp_grid(os.path.join(HERE, "openCARP", name))
    assert len(mg) >= 1 and all(t == kind for t in mg.cell_types)
    assert mg.nodes.shape[1] == 3
    check_orientation(mg)
    assert sorted(i for s in mg.cellsets.values() for i in s) == list(range(len(mg)))   # every element carries a region tag


@pytest.mark.parametrize("name,kind", CASES + [("ref-pyramid", "Pyramid")])
def test_mfem_reference_files(tb, name, kind):
    mg = tb.meshio.load_mfem_grid(os.path.join(HERE, "mfem", name + ".mesh"))
    assert len(mg) >= 1 and all(t == kind for t in mg.cell_types)
    check_orientation(mg)
    assert set(mg.cellsets) == {"1"}


def test_voom2_reference_file(tb):
    mg = tb.meshio.load_voom2_grid(os.path.join(HERE, "voom2", "ex1"))
    assert len(mg.nodes) == 9 and len(mg) == 2
    assert mg.cell_types == ["Line", "Hexahedron"]
    check_orientation(mg)
    with pytest.raises(ValueError):
        tb.meshio.load_mfem_grid(os.path.join(HERE, "voom2", "ex1.ele"))


def test_loaded_cells_integrate_like_the_oracle(tb, oracle):
    """ref-cube (volume 1) and ref-tetrahedron (volume 1/6) through the oracle: Σ M = volume, detJ > 0 everywhere."""
    for loader, path, okind, vol in ((tb.meshio.load_carp_grid, "openCARP/ref-cube", oracle.HEX8, 1.0),
                                     (tb.meshio.load_mfem_grid, "mfem/ref-cube.mesh", oracle.HEX8, 1.0),
                                     (tb.meshio.load_carp_grid, "openCARP/ref-tetrahedron", oracle.TET4, 1.0 / 6.0),
                                     (tb.meshio.load_mfem_grid, "mfem/ref-tetrahedron.mesh", oracle.TET4, 1.0 / 6.0)):
        g = loader(os.path.join(HERE, path)).grid()
        cd, nd = oracle.close_dofs(okind, 1, g.conn, g.n_nodes)
        rp, ci = oracle.build_pattern(cd, nd)
        om = oracle.Mesh(okind, 2, g.xyz, g.conn, cd)
        M = oracle.assemble_matrix(om, 0, oracle.Coef(oracle.COEF_CONST_SCALAR, [1.0]), rp, ci)
        assert M.sum() == pytest.approx(vol, rel=1e-14)
    g = tb.meshio.load_voom2_grid(os.path.join(HERE, "voom2", "ex1")).grid("Hexahedron")
    assert g.n_cells == 1 and list(g.file_cell_index) == [1]
    cd, nd = oracle.close_dofs(oracle.HEX8, 1, g.conn, g.n_nodes)
    om = oracle.Mesh(oracle.HEX8, 2, g.xyz, g.conn, cd)
    rp, ci = oracle.build_pattern(cd, nd)
    assert oracle.assemble_matrix(om, 0, oracle.Coef(oracle.COEF_CONST_SCALAR, [1.0]), rp, ci).sum() == pytest.approx(0.02 ** 3, rel=1e-10)


@pytest.mark.gpu
def test_loaded_meshes_assemble_on_device(tb, oracle, device):
    for loader, path, okind in ((tb.meshio.load_carp_grid, "openCARP/ref-cube", oracle.HEX8), (tb.meshio.load_mfem_grid, "mfem/ref-tetrahedron.mesh", oracle.TET4),
                                (tb.meshio.load_voom2_grid, "voom2/ex1", oracle.HEX8)):
        g = loader(os.path.join(HERE, path)).grid()
        cd, nd = oracle.close_dofs(okind, 1, g.conn, g.n_nodes)
        dh = tb.DofHandler(g, cell_dofs=cd, ndofs=nd)
        sp = tb.allocate_matrix(dh)
        om = oracle.Mesh(okind, 2, g.xyz, g.conn, cd)
        ref = oracle.assemble_matrix(om, 1, oracle.Coef(oracle.COEF_CONST_SCALAR, [1.0]), sp.rowptr, sp.colidx)
        for st in (tb.PatchAssemblyStrategy(device), tb.AtomicAssemblyStrategy(device)):
            K = tb.update_operator(tb.setup_operator(st, tb.BilinearDiffusionIntegrator(tb.ConstantCoefficient(1.0)), dh, sp), 0.0)
            assert np.abs(K.A.to_host() - ref).max() <= 1e-12 * np.abs(ref).max()
