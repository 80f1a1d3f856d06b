"""Whole-mesh pin of the oracle on a real Ferrite / Thunderbolt.jl run — when the maintainer-side fixtures exist.

`julia/make_golden.jl` (needs Julia + Thunderbolt.jl + Ferrite: not runnable in the build image) writes
tests/golden/ferrite_box_4x3x5.json and ferrite_quad_287x1.json: node / cell order of generate_grid, the dof table of
close!(dh), the CSR pattern as create_system_matrix makes it (src/solver/interface.jl:162-168), Ferrite's Gauss-point
order, and M / K / b assembled by Thunderbolt's own element routines in the sequential loop of
src/modeling/core/coordinate_systems.jl:145-171.  These tests compare the oracle with them entry by entry.  While the
files are absent every test here SKIPS with the word "unpinned": the conventions then rest on SURVEY §8(c)'s reading of
the Ferrite sources, exactly as before."""
import json
import os

import numpy as np
import pytest

HERE = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    path = os.path.join(HERE, name)
    if not os.path.exists(path):
        pytest.skip("unpinned: %s is absent — run julia/make_golden.jl where Thunderbolt.jl + Ferrite are installed" % name)
    return json.load(open(path))


def _perturb(xyz, nel, amp):
    """tb_host_perturb_nodes (thunderbolt.jl_amd/csrc/tb_hostgen.cpp), restated"""
    nx, ny, nz = nel
    px, py = nx + 1, ny + 1
    idx = np.arange(len(xyz))
    i, j, k = idx % px, (idx // px) % py, idx // (px * py)
    h = (xyz[-1] - xyz[0]) / np.array(nel)
    s = np.sin(2 * np.pi * i / nx) * np.sin(2 * np.pi * j / ny) * np.sin(2 * np.pi * k / nz)
    out = xyz.copy()
    out[:, 0] += amp * h[0] * s
    out[:, 1] -= 0.5 * amp * h[1] * s
    out[:, 2] += 0.75 * amp * h[2] * s
    return out


def test_box_mesh_dofs_and_pattern_are_ferrites(oracle):
    g = _load("ferrite_box_4x3x5.json")
    o = oracle
    nel = tuple(g["nel"])
    xyz, conn = o.generate_grid_hex(*nel, (0, 0, 0), (1, 1, 1))
    xyz = _perturb(xyz, nel, g["perturb"])
    np.testing.assert_allclose(xyz, np.asarray(g["nodes"]), rtol=0, atol=1e-15)          # node order + corner interpolation
    np.testing.assert_array_equal(conn, np.asarray(g["cells"]))                          # cell order + local vertex order
    cd, nd = o.close_dofs(o.HEX8, 1, conn, len(xyz))
    assert nd == g["ndofs"]
    np.testing.assert_array_equal(cd, np.asarray(g["celldofs"]))                         # close!(dh) numbering
    rp, ci = o.build_pattern(cd, nd)
    np.testing.assert_array_equal(rp, np.asarray(g["rowptr"]))
    np.testing.assert_array_equal(ci, np.asarray(g["colval"]))                           # sorted columns per row


def test_gauss_point_order_is_ferrites(oracle):
    g = _load("ferrite_box_4x3x5.json")
    xi, w = oracle.quadrature(oracle.HEX8, 2)
    np.testing.assert_allclose(xi, np.asarray(g["gauss_points"]), atol=1e-15)
    np.testing.assert_allclose(w, np.asarray(g["gauss_weights"]), atol=1e-15)


def test_box_matrices_and_linear_form_equal_thunderbolts(oracle):
    g = _load("ferrite_box_4x3x5.json")
    o = oracle
    xyz, conn, cd = np.asarray(g["nodes"], dtype=float), np.asarray(g["cells"], dtype=np.int32), np.asarray(g["celldofs"], dtype=np.int32)
    rp, ci = np.asarray(g["rowptr"], dtype=np.int64), np.asarray(g["colval"], dtype=np.int32)   # Ferrite's own tables: values are compared on identical indexing
    m = o.Mesh(o.HEX8, 2, xyz, conn, cd)
    M = o.assemble_matrix(m, 0, o.Coef(o.COEF_CONST_SCALAR, [1.0]), rp, ci)
    K = o.assemble_matrix(m, 1, o.Coef(o.COEF_CONST_TENSOR, g["D"]), rp, ci)
    b = o.assemble_source(m, o.SRC_NORM_PLUS_T, t=0.0)
    for got, key in ((M, "M"), (K, "K"), (b, "b_norm_x_plus_t")):
        ref = np.asarray(g[key])
        assert np.abs(got - ref).max() <= 1e-10 * np.abs(ref).max(), key                       # north_star's bar; only summation order may differ


def test_quad_287x1_linear_form_equals_thunderbolts(oracle):
    """mesh and source of the reference's GPU operator test (test/gpu/test_operators.jl:1-31)"""
    g = _load("ferrite_quad_287x1.json")
    o = oracle
    xyz2 = np.ascontiguousarray(np.asarray(g["nodes"], dtype=float))                          # the oracle's quadrilaterals live in the plane
    conn, cd = np.asarray(g["cells"], dtype=np.int32), np.asarray(g["celldofs"], dtype=np.int32)
    nx, ny = g["nel"]
    px = nx + 1
    lattice = np.stack(np.meshgrid(np.linspace(-1, 1, px), np.linspace(-1, 1, ny + 1), indexing="xy"), axis=-1).reshape(-1, 2)
    np.testing.assert_allclose(xyz2, lattice, atol=1e-15)                                      # x fastest
    c = np.arange(nx * ny)
    n0 = (c % nx) + (c // nx) * px
    np.testing.assert_array_equal(conn, np.stack([n0, n0 + 1, n0 + 1 + px, n0 + px], axis=1))
    cd_o, nd = o.close_dofs(o.QUAD4, 1, conn, len(xyz2))
    assert nd == g["ndofs"]
    np.testing.assert_array_equal(cd_o, cd)
    rp, ci = o.build_pattern(cd, nd)
    np.testing.assert_array_equal(rp, np.asarray(g["rowptr"]))
    np.testing.assert_array_equal(ci, np.asarray(g["colval"]))
    b = o.assemble_source(o.Mesh(o.QUAD4, 2, xyz2, conn, cd), o.SRC_COS_EXP, t=0.0)
    ref = np.asarray(g["b_cos_exp"])
    assert np.abs(b - ref).max() <= 1e-10 * np.abs(ref).max()
