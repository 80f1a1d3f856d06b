#!/usr/bin/env python3
"""BASELINE config 5 at toy size, one MI355X: coupled electromechanics on the idealised left ventricle (all-hexahedral, O-grid apex).
Electrophysiology — monodomain with the ten Tusscher–Panfilov 2006 ionic model, fibre-aligned conductivities from the rule-based fibre
field, operator splitting (backward-Euler diffusion by Jacobi-CG + Rush–Larsen reaction), apical stimulus — drives the mechanics through
the intracellular calcium of every node: ActiveStressModel(Holzapfel–Ogden 2009, SimpleActiveStress, CaDrivenInternalSarcomereModel(
PelceSunLangeveld1995, Ca(x, t))) with the reference's ventricle boundary conditions (epicardial Robin bed, basal normal springs,
endocardial pressure, four anchors), one Newton solve every `--mech-every` ms.  The coupling is one-way (calcium → tension), like the
reference's contraction tutorials.  Prints one JSON line with the wave front, calcium range, apex shortening and timings."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--nc", type=int, default=16); ap.add_argument("--nr", type=int, default=3); ap.add_argument("--nl", type=int, default=8)
ap.add_argument("--tend", type=float, default=120.0, help="ms")
ap.add_argument("--dt", type=float, default=0.05, help="electrophysiology time step [ms]")
ap.add_argument("--mech-every", type=float, default=5.0, help="mechanics solve interval [ms]")
ap.add_argument("--tmax", type=float, default=20.0, help="peak active tension")
# the follower pressure makes the tangent non-symmetric: the Newton steps are solved by restarted GMRES with a Jacobi preconditioner on the device
args = ap.parse_args()
import thunderbolt_jl_amd as tb
dev = tb.MI355XDevice(0)
t0 = time.perf_counter()
g = tb.generate_ideal_lv_mesh_hex(args.nc, args.nr, args.nl)
f, s, n = tb.ideal_lv_microstructure(g, np.deg2rad(60.0), np.deg2rad(-60.0))
# ---- electrophysiology on the scalar Q1 field
dhs = tb.DofHandler(g)
sps = tb.allocate_matrix(dhs)
lam = np.array([0.3, 0.12, 0.12]) * 1e-2                                   # conductivities along f, s, n (mm²/ms scale of this unit ventricle)
D = tb.ConductivityToDiffusivityCoefficient(tb.SpectralTensorCoefficient(tb.OrthotropicMicrostructureModel(f, s, n), tb.ConstantCoefficient(lam)),
                                            tb.ConstantCoefficient(1.0), tb.ConstantCoefficient(1.0))
heat = tb.BackwardEulerStage(tb.BackwardEulerSolver(rtol=1e-6, atol=1e-8), tb.PerColorAssemblyStrategy(dev), dhs, D, None, sps)
model = tb.TT06()
npts = dhs.ndofs
n2d = tb.distributed.node_to_dof(dhs)
X = np.empty((npts, 3)); X[n2d] = g.xyz
u0 = np.tile(model.default_initial_state(), (npts, 1)).T.copy()
apex = g.xyz[g.getnodeset("Apex")[0]]
u0[model.phi_index, np.linalg.norm(X - apex, axis=1) < 0.45] = 20.0         # depolarised apical region
fode = tb.PointwiseODEFunction(npts, model)
cache = tb.setup_solver_cache(fode, tb.RushLarsenCellSolver(dev), u=dev.to_device(np.ascontiguousarray(u0).ravel()), keep_du=False)
ltg = tb.LieTrotterGodunov(heat, fode, cache)
ica = model.state_symbols.index("Ca_i")
ca_rest = float(model.default_initial_state()[ica])
# ---- mechanics on the vector Q1 field
dhv = tb.DofHandler(g, tb.LagrangeCollection(1) ** 3)
spv = tb.allocate_matrix(dhv)
calcium = {"field": np.zeros((g.n_cells, 8))}
sarc = tb.CaDrivenInternalSarcomereModel(tb.PelceSunLangeveld1995Model(), lambda t: calcium["field"])
cm = tb.ActiveStressModel(tb.HolzapfelOgden2009Model(), tb.SimpleActiveStress(Tmax=args.tmax), sarc, tb.OrthotropicMicrostructureModel(f, s, n))
face = (tb.RobinBC(0.1, "Epicardium"), tb.NormalSpringBC(1.0, "Base"), tb.PressureFieldBC(tb.ConstantCoefficient(0.01), "Endocardium"))
op = tb.setup_operator(tb.ElementAssemblyStrategy(dev), tb.QuasiStaticModel("d", cm, face), dhv, spv)
nd0 = np.empty(g.n_nodes, dtype=np.int64)
nd0[g.conn.ravel()] = dhv.cell_dofs[:, 0::3].ravel()
a = [g.getnodeset("MyocardialAnchor%d" % k)[0] for k in (1, 2, 3, 4)]
ch = tb.ConstraintHandler(dhv, np.concatenate([nd0[a[0]] + np.arange(3), nd0[a[1]] + np.array([1, 2]), [nd0[a[2]] + 2], [nd0[a[3]] + 2]]))
# the first increment after a change of the calcium field may raise the residual once before quadratic convergence takes over (the reference
# makes the same remark about its condensed cuboid): no monotonicity enforcement
newton = tb.NewtonRaphsonSolver(max_iter=30, tol=1e-8, inner_rtol=1e-10, inner_maxiter=20000, inner_solver="gmres",
                                gmres_restart=200, enforce_monotonic_convergence=False)
u = dev.zeros(dhv.ndofs)
# passive inflation to the cavity pressure before the beat starts (calcium at rest): a short load path from the unloaded state
face[2].param_of_t = lambda tt: 0.01 * min(1.0, tt)
assert tb.HomotopyPathSolver(newton).solve(u, op, ch, (0.0, 1.0), 0.25, adaptive=True), "passive inflation failed"
face[2].param_of_t = None
setup_s = time.perf_counter() - t0
# ---- time loop
nsteps = int(round(args.tend / args.dt))
every = max(1, int(round(args.mech_every / args.dt)))
ep_s = mech_s = 0.0
hist = []
apex_dof_z = nd0[g.getnodeset("Apex")[0]] + 2
for k in range(nsteps):
    t = k * args.dt
    t1 = time.perf_counter()
    assert ltg.step(t, args.dt)
    if (k + 1) % every == 0:
        dev.synchronize()
        ep_s += time.perf_counter() - t1
        t1 = time.perf_counter()
        un = cache.un.to_host().reshape(model.nstates, npts)
        ca_node = np.empty(g.n_nodes); ca_node = un[ica][n2d]
        # normalised calcium transient in [0, 1]: rest → 0, systolic peak (≈ 1 µM for TT06) → 1
        ca_new = np.clip((ca_node[g.conn] - ca_rest) / (1.0e-3 - ca_rest), 0.0, 1.0)
        # the calcium field is the load parameter of this solve: walk from the previous field to the new one, halving the increment when Newton
        # fails (a coarse mesh does not take a large activation jump in one step)
        ca_old, frac, inc, ok = calcium["field"].copy(), 0.0, 1.0, True
        while frac < 1.0 - 1e-12:
            trial = min(1.0, frac + inc)
            calcium["field"] = ca_old + trial * (ca_new - ca_old)
            ub = u.to_host()
            if tb.nlsolve(u, op, ch, newton, t=t + args.dt):
                frac = trial
            else:
                u.copy_from_host(ub)
                inc *= 0.5
                if inc < 1.0 / 64:
                    ok = False
                    break
        dev.synchronize()
        mech_s += time.perf_counter() - t1
        phi = un[model.phi_index]
        hist.append({"t": t + args.dt, "activated_fraction": float((phi > -40.0).mean()), "ca_max_uM": float(ca_node.max() * 1e3),
                     "apex_uz": float(u.to_host()[apex_dof_z]), "newton_iters": newton.iter, "residual": newton.residual_norms[-1], "ok": bool(ok)})
        if not ok:
            hist[-1]["residual_norms"] = newton.residual_norms
            hist[-1]["theta"] = [float(x) for x in newton.theta]
            hist[-1]["linear_failure"] = newton.linear_failure
        if not ok:
            break
    else:
        ep_s += time.perf_counter() - t1
print(json.dumps({"workload": "electromechanics on the ideal LV: %d hexahedra, %d EP dofs × %d TT06 states, %d displacement dofs; EP dt %g ms, mechanics every %g ms"
                              % (g.n_cells, npts, model.nstates, dhv.ndofs, args.dt, args.mech_every),
                  "setup_s": setup_s, "ep_s": ep_s, "ep_ms_per_step": ep_s / nsteps * 1e3, "mechanics_s": mech_s, "mechanics_solves": len(hist),
                  "all_converged": all(h["ok"] for h in hist), "history": hist[:: max(1, len(hist) // 8)], "final": hist[-1] if hist else None}))
