#!/usr/bin/env python3
"""Land et al. 2015, cardiac mechanics verification benchmark, problem 1 (the reference's test/validation/land2015.jl) on one MI355X:
a 10×1×1 Guccione beam clamped at x = 0 under a follower pressure of 0.004 on its bottom face, quadratic hexahedra, load path in steps of
0.2 with Newton; reports the z-deflection of the point (10, 0.5, 1) — 3.17 ± 0.02 in the reference's assertion — and timings.
--refine r uses (25r)×(3r)×(3r) cells; --inner picks the linear solver of the Newton steps, all on the device: CG with the Chebyshev polynomial
preconditioner (default: the slender beam is badly conditioned — Jacobi-CG needs thousands of iterations per Newton step, degree-24 Chebyshev
about twenty times fewer), Jacobi-CG, or GMRES(200), which stagnates on this bending problem with a Jacobi preconditioner and is kept only to show it."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--refine", type=int, default=1)
ap.add_argument("--inner", default="chebyshev", choices=["chebyshev", "cg", "gmres"])
args = ap.parse_args()
import thunderbolt_jl_amd as tb
dev = tb.MI355XDevice(0)
r = args.refine
t0 = time.perf_counter()
g = tb.generate_mesh(tb.Hexahedron, (25 * r, 3 * r, 3 * r), (0.0, 0.0, 0.0), (10.0, 1.0, 1.0))
dh = tb.DofHandler(g, tb.LagrangeCollection(2) ** 3)
sp = tb.allocate_matrix(dh)
mat = tb.Guccione1991PassiveModel(C0=2.0, Bff=8.0, Bss=2.0, Bnn=2.0, Bns=1.0, Bfs=2.0, Bfn=2.0, mpU=tb.SimpleCompressionPenalty(100.0))
ms = tb.ConstantCoefficient(tb.OrthotropicMicrostructure([1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0]))
load = (tb.PressureFieldBC(lambda t: min(t, 1.0) * 0.004, "bottom"),)
op = tb.setup_operator(tb.ElementAssemblyStrategy(dev), tb.QuasiStaticModel("displacement", tb.PK1Model(mat, ms), load), dh, sp)
sgn = np.array([[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1]], dtype=float)
tix = np.array([(0, 0, 0), (2, 0, 0), (2, 2, 0), (0, 2, 0), (0, 0, 2), (2, 0, 2), (2, 2, 2), (0, 2, 2), (1, 0, 0), (2, 1, 0), (1, 2, 0), (0, 1, 0), (1, 0, 2),
                (2, 1, 2), (1, 2, 2), (0, 1, 2), (0, 0, 1), (2, 0, 1), (2, 2, 1), (0, 2, 1), (1, 1, 0), (1, 0, 1), (2, 1, 1), (1, 2, 1), (0, 1, 1), (1, 1, 2), (1, 1, 1)],
               dtype=float) - 1.0
N = 0.125 * np.prod(1.0 + sgn[None, :, :] * tix[:, None, :], axis=2)
pos = np.einsum("ba,cak->cbk", N, g.xyz[g.conn])
X = np.empty((dh.ndofs, 3))
for c in range(3):
    X[dh.cell_dofs[:, c::3].ravel()] = pos.reshape(-1, 3)
ch = tb.ConstraintHandler(dh, np.flatnonzero(X[:, 0] < 1e-12))
setup_s = time.perf_counter() - t0


if args.inner == "chebyshev":
    inner, precond = "cg", tb.ChebyshevPrecBuilder(24)
else:
    inner, precond = args.inner, None
newton = tb.NewtonRaphsonSolver(tol=1e-4, max_iter=10, inner_solver=inner, inner_precond=precond, inner_rtol=1e-8, inner_maxiter=20000, gmres_restart=200)
path = tb.HomotopyPathSolver(newton)
u = dev.zeros(dh.ndofs)
dev.synchronize()
t0 = time.perf_counter()
t, ok, newton_its, linear_its = 0.0, True, 0, 0
while ok and t < 1.0 - 1e-12:
    h = min(0.2, 1.0 - t)
    ok = path.solve(u, op, ch, (t, t + h), h, adaptive=True, maxiters=100)
    newton_its += sum(s[2] for s in path.steps)
    t += h
dev.synchronize()
solve_s = time.perf_counter() - t0
uh = u.to_host()
zset = set(dh.cell_dofs[:, 2::3].ravel().tolist())
tip = [d for d in np.flatnonzero((np.abs(X[:, 0] - 10.0) < 1e-9) & (np.abs(X[:, 1] - 0.5) < 1e-9) & (np.abs(X[:, 2] - 1.0) < 1e-9)) if d in zset]
print(json.dumps({"workload": "Land 2015 problem 1, Q2, %dx%dx%d cells, %d dofs, inner solver %s" % (25 * r, 3 * r, 3 * r, dh.ndofs, args.inner),
                  "converged": bool(ok), "tip_deflection_z": float(uh[tip[0]]) if tip else None, "reference_assertion": "3.17 +- 0.02",
                  "newton_iterations": int(newton_its), "setup_s": setup_s, "solve_s": solve_s}))
