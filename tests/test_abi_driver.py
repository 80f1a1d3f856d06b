"""The C ABI driven by a plain C program (tests/abi_driver.c: no Python, no ctypes) in the reference's call order — set-up of the heat stage,
then per time step perform_backward_euler_step! (src/solver/time/euler.jl:71-101) and the pointwise cell step
(src/solver/time/partitioned_solver.jl:38-52) — against the same steps done by the CPU oracle."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, EXE = os.path.join(ROOT, "tests", "abi_driver.c"), os.path.join(ROOT, "tests", "abi_driver")
LIBDIR = os.path.join(ROOT, "thunderbolt.jl_amd")


def _build():
    cmd = ["gcc", "-O2", "-std=c11", "-Wall", "-I", os.path.join(ROOT, "include"), SRC, "-L", LIBDIR, "-ltbhip", "-lm",
           "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib", "-o", EXE]
    subprocess.check_call(cmd)


def test_abi_driver_compiles_against_the_header():
    """a C11 compiler accepts include/tbhip.h and links every entry point the driver uses (no GPU needed)"""
    _build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_abi_driver_matches_the_oracle(oracle):
    import scipy.sparse as sps
    import scipy.sparse.linalg as spla
    o = oracle
    _build()
    n, nsteps, dt = 16, 5, 0.5
    r = subprocess.run([EXE, str(n), str(nsteps)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    head = lines[0].split()
    got = np.array([float.fromhex(x) for x in lines[1:]])
    xyz, conn = o.generate_grid_hex(n, n, n, (0, 0, 0), (1, 1, 1))
    # the driver's smooth distortion (tb_host_perturb_nodes: x += 0.2 h s, y −= 0.1 h s, z += 0.15 h s, s = Π sin(2π i_d / n)) restated here
    i = np.arange(n + 1)
    sx = np.sin(2 * np.pi * i / n)
    s3 = (sx[None, None, :] * sx[None, :, None] * sx[:, None, None]).ravel()          # node id = i + (n+1)(j + (n+1)k)
    h = 1.0 / n
    xyz = xyz + 0.2 * h * s3[:, None] * np.array([1.0, -0.5, 0.75])
    cd, nd = o.close_dofs(o.HEX8, 1, conn, len(xyz))
    rp, ci = o.build_pattern(cd, nd)
    assert int(head[1]) == nd and int(head[3]) == len(ci)
    om = o.Mesh(o.HEX8, 2, xyz, conn, cd)
    kap = np.array([[4.5e-3, 5.0e-4, 0], [5.0e-4, 2.0e-3, 0], [0, 0, 2.0e-3]])
    M = o.assemble_matrix(om, 0, o.Coef(o.COEF_CONST_SCALAR, [1.0]), rp, ci)
    K = o.assemble_matrix(om, 1, o.Coef(o.COEF_CONST_TENSOR, kap.ravel(), Cm=1.0, chi=2.0, wrap=True), rp, ci)
    A = sps.csr_matrix((o.heat_matrix(M, K, dt), ci, rp), shape=(nd, nd))
    Ms = sps.csr_matrix((M, ci, rp), shape=(nd, nd))
    lu = spla.splu(A.tocsc())
    u = np.zeros(2 * nd)
    X = xyz[conn.ravel()]
    d = cd.ravel()
    u[d] = ((X[:, 0] <= 0.5) & (X[:, 1] <= 0.5)).astype(float)
    u[nd + d] = np.where(X[:, 1] >= 0.5, 0.1, 0.0)
    p = o.cell_default_params(o.CELL_FHN)
    t = 0.0
    for _ in range(nsteps):
        b = Ms @ u[:nd] + o.assemble_source(om, o.SRC_COS_EXP, t=t + dt)
        u[:nd] = lu.solve(b)
        o.reaction_step(o.CELL_FHN, p, u, nd, o.LAYOUT_SOA, t=t, dt=dt, want_du=False)
        t += dt
    assert np.abs(got - u).max() < 1e-8 * np.abs(u).max()
