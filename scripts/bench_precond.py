#!/usr/bin/env python3
"""Jacobi vs ℓ₁ Gauss–Seidel vs Chebyshev-polynomial preconditioned CG on the device: the heat matrix M − Δt·K of an n³ mesh and a Q2 mechanics tangent."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=100)
ap.add_argument("--dt", type=float, default=1.0)
ap.add_argument("--mech-n", type=int, default=16)
args = ap.parse_args()
import thunderbolt_jl_amd as tb
dev = tb.MI355XDevice(0)
out = {}
g = tb.generate_mesh(tb.Hexahedron, (args.n,) * 3, (0, 0, 0), (1.0, 1.0, 1.0), perturb=0.1)
dh = tb.DofHandler(g)
sp = tb.allocate_matrix(dh)
st = tb.PatchAssemblyStrategy(dev)
kap = np.diag([4.5e-5, 2.0e-5, 2.0e-5]) * 10
D = tb.ConductivityToDiffusivityCoefficient(tb.ConstantCoefficient(kap), tb.ConstantCoefficient(1.0), tb.ConstantCoefficient(1.0))
M = tb.update_operator(tb.setup_operator(st, tb.BilinearMassIntegrator(tb.ConstantCoefficient(1.0)), dh, sp), 0.0)
K = tb.update_operator(tb.setup_operator(st, tb.BilinearDiffusionIntegrator(D), dh, sp), 0.0)
A = dev.zeros(sp.nnz)
tb.check(tb.lib().tb_heat_matrix(dev.h, sp.nnz, M.A.ptr, K.A.ptr, args.dt, A.ptr))
rng = np.random.default_rng(0)
b = dev.to_device(rng.normal(size=dh.ndofs))
pat = M.pattern
for name, pc in (("jacobi", "jacobi"), ("l1gs_64", tb.L1GSPrecBuilder(64)), ("chebyshev_4", tb.ChebyshevPrecBuilder(4)), ("chebyshev_8", tb.ChebyshevPrecBuilder(8))):
    x = dev.zeros(dh.ndofs)
    tb.pcg_solve(pat, A, b, x, rtol=1e-8, atol=0.0, maxiter=3, precond=pc)
    x = dev.zeros(dh.ndofs); dev.synchronize()
    t0 = time.perf_counter()
    its, res = tb.pcg_solve(pat, A, b, x, rtol=1e-8, atol=0.0, maxiter=5000, precond=pc)
    dev.synchronize()
    out["heat_%d^3_%s" % (args.n, name)] = {"iterations": its, "ms": (time.perf_counter() - t0) * 1e3}
n = args.mech_n
gm = tb.generate_mesh(tb.Hexahedron, (n, n, n), (0, 0, 0), (1.0, 1.0, 1.0))
dhm = tb.DofHandler(gm, tb.LagrangeCollection(2) ** 3)
spm = tb.allocate_matrix(dhm)
fsn = tb.ConstantCoefficient(tb.OrthotropicMicrostructure([1, 0, 0], [0, 1, 0], [0, 0, 1]))
op = tb.setup_operator(tb.ElementAssemblyStrategy(dev), tb.QuasiStaticModel("u", tb.PK1Model(tb.HolzapfelOgden2009Model(), fsn)), dhm, spm)
u = dev.zeros(dhm.ndofs)
r = dev.zeros(dhm.ndofs)
tb.update_linearization(op, u, 0.0, residual=r)
# clamp the x = 0 face: dofs of nodes with x = 0 (positions through the cell dof table of the corner nodes suffices for Q1 corners; use all dofs on the plane via a Q2 position table)
X = np.zeros((dhm.ndofs, 3))
sgn = np.array([[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1]], dtype=float)
tix = np.array([(0, 0, 0), (2, 0, 0), (2, 2, 0), (0, 2, 0), (0, 0, 2), (2, 0, 2), (2, 2, 2), (0, 2, 2), (1, 0, 0), (2, 1, 0), (1, 2, 0), (0, 1, 0), (1, 0, 2), (2, 1, 2), (1, 2, 2),
                (0, 1, 2), (0, 0, 1), (2, 0, 1), (2, 2, 1), (0, 2, 1), (1, 1, 0), (1, 0, 1), (2, 1, 1), (1, 2, 1), (0, 1, 1), (1, 1, 2), (1, 1, 1)], dtype=float) - 1.0
N = 0.125 * np.prod(1.0 + sgn[None, :, :] * tix[:, None, :], axis=2)
pos = np.einsum("ba,cak->cbk", N, gm.xyz[gm.conn])
for c in range(3):
    X[dhm.cell_dofs[:, c::3].ravel()] = pos.reshape(-1, 3)
ch = tb.ConstraintHandler(dhm, np.flatnonzero(X[:, 0] < 1e-12))
tb.apply_zero(op.J, r, ch, pattern=op.pattern)
bm = dev.to_device(rng.normal(size=dhm.ndofs) * (X[:, 0] > 1e-12))
for name, pc in (("jacobi", "jacobi"), ("l1gs_96", tb.L1GSPrecBuilder(96)), ("chebyshev_8", tb.ChebyshevPrecBuilder(8)), ("chebyshev_16", tb.ChebyshevPrecBuilder(16)),
                 ("chebyshev_24", tb.ChebyshevPrecBuilder(24))):
    x = dev.zeros(dhm.ndofs); dev.synchronize()
    t0 = time.perf_counter()
    its, res = tb.pcg_solve(op.pattern, op.J, bm, x, rtol=1e-8, atol=0.0, maxiter=6000, precond=pc)
    dev.synchronize()
    out["mechanics_Q2_%d^3_%s" % (n, name)] = {"iterations": its, "ms": (time.perf_counter() - t0) * 1e3, "dofs": dhm.ndofs}
print(json.dumps(out))
