#!/usr/bin/env python3
"""Quasi-static contraction of a myocardial block on one MI355X: Holzapfel–Ogden 2009 passive tissue with fibres along x,
active stress Ta(t) = Tmax·Ca(t) (ActiveStressModel + SimpleActiveStress), the x = 0 face clamped, a Robin spring bed on
the x = L face (RobinBC), solved by load stepping in time with Newton–Raphson (device tangent assembly on the FP64 matrix
cores, device Dirichlet elimination, Jacobi-CG).  The device analogue of the reference's contracting-cuboid setups
(docs/src/literate-tutorials/cm01_simple-active-stress.jl, test/integration/test_solid_mechanics.jl).  Prints one JSON line."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=12)
ap.add_argument("--order", type=int, default=2)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--tmax", type=float, default=0.5, help="peak active tension")
ap.add_argument("--sarcomere", default="steady", choices=["steady", "rdq20", "rdq20-rate"],
                help="steady: Ta = Tmax·Ca(t); rdq20: RDQ20-MF internal state condensed per quadrature point (AsRateIndependent); rdq20-rate: rate-coupled")
ap.add_argument("--dt", type=float, default=2.0, help="time step of the internal variables [ms] (rdq20 variants)")
args = ap.parse_args()
import thunderbolt_jl_amd as tb
dev = tb.MI355XDevice(0)
n = args.n
g = tb.generate_mesh(tb.Hexahedron, (n, n, n), (0, 0, 0), (1.0, 1.0, 1.0))
dh = tb.DofHandler(g, tb.LagrangeCollection(args.order) ** 3)
sp = tb.allocate_matrix(dh)
calcium = lambda t: np.sin(0.5 * np.pi * min(t, 1.0)) ** 2                      # noqa: E731  (0 → 1 over the load steps)
fsn = tb.ConstantCoefficient(tb.OrthotropicMicrostructure([1, 0, 0], [0, 1, 0], [0, 0, 1]))
material = tb.ActiveStressModel(tb.HolzapfelOgden2009Model(), tb.SimpleActiveStress(Tmax=args.tmax),
                                tb.CaDrivenInternalSarcomereModel(tb.PelceSunLangeveld1995Model(), calcium), fsn)
if args.sarcomere != "steady":
    # calcium transient of the reference's sarcomere test (test/test_sarcomere.jl:25-35), time in ms; Tmax scales the cross-bridge tension
    c0, cmax, tau1, tau2, tstart = 0.1, 0.9, 20.0, 50.0, 10.0
    beta = (tau1 / tau2) ** (-1 / (tau1 / tau2 - 1)) - (tau1 / tau2) ** (-1 / (1 - tau2 / tau1))
    calcium = lambda t: c0 if t < tstart else c0 + (cmax - c0) / beta * (np.exp(-(t - tstart) / tau1) - np.exp(-(t - tstart) / tau2))   # noqa: E731
    sm = tb.RDQ20MFModel()
    material = tb.ActiveStressModel(tb.HolzapfelOgden2009Model(), tb.SimpleActiveStress(Tmax=args.tmax),
                                    tb.CaDrivenInternalSarcomereModel(sm if args.sarcomere == "rdq20-rate" else tb.AsRateIndependent(sm), calcium), fsn)
model = tb.QuasiStaticModel("u", material, [tb.RobinBC(0.05, "right")])
t0 = time.perf_counter()
op = tb.setup_operator(tb.ElementAssemblyStrategy(dev), model, dh, sp)
# clamp x = 0: dofs of the nodes on that plane (positions of the Q1/Q2 nodes through the trilinear map)
sgn = np.array([[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1]], dtype=float)
tix = np.array([(0, 0, 0), (2, 0, 0), (2, 2, 0), (0, 2, 0), (0, 0, 2), (2, 0, 2), (2, 2, 2), (0, 2, 2), (1, 0, 0), (2, 1, 0), (1, 2, 0), (0, 1, 0), (1, 0, 2),
                (2, 1, 2), (1, 2, 2), (0, 1, 2), (0, 0, 1), (2, 0, 1), (2, 2, 1), (0, 2, 1), (1, 1, 0), (1, 0, 1), (2, 1, 1), (1, 2, 1), (0, 1, 1), (1, 1, 2), (1, 1, 1)], dtype=float) - 1.0
xi = sgn if args.order == 1 else tix
N = 0.125 * np.prod(1.0 + sgn[None, :, :] * xi[:, None, :], axis=2)
pos = np.einsum("ba,cak->cbk", N, g.xyz[g.conn])
X = np.empty((dh.ndofs, 3))
for c in range(3):
    X[dh.cell_dofs[:, c::3].ravel()] = pos.reshape(-1, 3)
ch = tb.ConstraintHandler(dh, np.flatnonzero(X[:, 0] < 1e-12))
u = dev.zeros(dh.ndofs)
dev.synchronize()
t_setup = time.perf_counter() - t0
solver = tb.NewtonRaphsonSolver(max_iter=25, tol=1e-8, inner_rtol=1e-10, inner_solver="gmres" if args.sarcomere == "rdq20-rate" else "cg")
hist = []
t0 = time.perf_counter()
for s in range(1, args.steps + 1):
    if args.sarcomere == "steady":
        t = s / args.steps
        ok = tb.nlsolve(u, op, ch, solver, t=t)
        hist.append({"t": t, "Ta": args.tmax * calcium(t), "newton_iters": solver.iter, "cg_iters": int(sum(solver.linear_iters)), "residual": solver.residual_norms[-1], "ok": bool(ok)})
    else:   # backward Euler on the internal states, multi-level Newton (local problems re-solved inside every assembly)
        t = (s - 1) * args.dt
        ok = tb.perform_mechanics_step(u, op, ch, solver, t, args.dt)
        Q = op.internal.to_host()
        hist.append({"t": t + args.dt, "Ca": float(calcium(t + args.dt)), "newton_iters": solver.iter, "linear_iters": int(sum(solver.linear_iters)),
                     "residual": solver.residual_norms[-1], "mean_xb_force_state": float((Q[17] + Q[19]).mean()), "ok": bool(ok)})
    if not ok:
        break
dev.synchronize()
elapsed = time.perf_counter() - t0
uh = u.to_host()
xdofs = np.zeros(dh.ndofs, dtype=bool)
xdofs[dh.cell_dofs[:, 0::3].ravel()] = True                                       # x-components
ux_right = uh[xdofs & (X[:, 0] > 1 - 1e-12)]
print(json.dumps({"workload": "contracting block, HO2009 + active stress (%s), Q%d, %d^3 cells, %d dofs" % (args.sarcomere, args.order, n, dh.ndofs),
                  "setup_s": t_setup, "solve_s": elapsed, "steps": hist, "mean_shortening_x": float(-ux_right.mean()),
                  "all_converged": all(h["ok"] for h in hist)}))
