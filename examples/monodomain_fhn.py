#!/usr/bin/env python3
"""BASELINE config 2: monodomain + FitzHugh–Nagumo, n³-hex Q1 mesh, operator splitting
LieTrotterGodunov((BackwardEulerSolver(CG), ForwardEulerCellSolver())) on one MI355X — the device analogue of
docs/src/literate-tutorials/ep01_spiral-wave.jl (3-D box instead of the 2-D sheet).  Prints one JSON line."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=100)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--dt", type=float, default=None, help="time step (default 1.0 for FHN, 0.02 ms for TT06)")
ap.add_argument("--ionic", default="fhn", choices=["fhn", "tt06"], help="tt06: BASELINE config 3 (ten Tusscher 2006)")
ap.add_argument("--per-step", action="store_true", help="synchronise and print the wall time of every step (diagnostic)")
ap.add_argument("--cell-solver", default="rl", choices=["rl", "fe"], help="TT06: Rush-Larsen (one evaluation per step) or forward Euler with 20 sub-steps")
args = ap.parse_args()
import thunderbolt_jl_amd as tb
dev = tb.MI355XDevice(0)
L = 2.5
g = tb.generate_mesh(tb.Hexahedron, (args.n,) * 3, (0, 0, 0), (L, L, L))
dh = tb.DofHandler(g)
sp = tb.allocate_matrix(dh)
kap = np.diag([4.5e-5, 2.0e-5, 2.0e-5])                   # ep01_spiral-wave.jl:39-41
D = tb.ConductivityToDiffusivityCoefficient(tb.ConstantCoefficient(kap), tb.ConstantCoefficient(1.0), tb.ConstantCoefficient(1.0))
t0 = time.perf_counter()
heat = tb.BackwardEulerStage(tb.BackwardEulerSolver(rtol=1e-5, atol=1e-6), tb.PatchAssemblyStrategy(dev), dh, D, None, sp)
dev.synchronize()
t_setup = time.perf_counter() - t0
model = tb.FHNModel() if args.ionic == "fhn" else tb.TT06()
if args.dt is None:
    args.dt = 1.0 if args.ionic == "fhn" else 0.02
n = dh.ndofs
n2d = tb.distributed.node_to_dof(dh)
X = np.empty((n, 3)); X[n2d] = g.xyz
if args.ionic == "fhn":
    u0 = np.zeros((2, n))
    u0[0] = ((X[:, 0] <= L / 2) & (X[:, 1] <= L / 2)).astype(float)   # ep01:113-118
    u0[1] = 0.1 * (X[:, 1] >= L / 2)
    cell_solver = tb.ForwardEulerCellSolver(dev)
else:  # resting tissue with a depolarised corner; the fast sodium gate needs Δt ≈ 1 µs-scale sub-steps under forward Euler
    u0 = np.tile(model.default_initial_state(), (n, 1)).T.copy()
    u0[model.phi_index, (X[:, 0] <= L / 4) & (X[:, 1] <= L / 4)] = 20.0
    cell_solver = tb.RushLarsenCellSolver(dev) if args.cell_solver == "rl" else tb.AdaptiveForwardEulerSubstepper(dev, substeps=20, reaction_threshold=0.0)
f = tb.PointwiseODEFunction(n, model)
cache = tb.setup_solver_cache(f, cell_solver, u=dev.to_device(np.ascontiguousarray(u0).ravel()), keep_du=False)
ltg = tb.LieTrotterGodunov(heat, f, cache)
ltg.step(0.0, args.dt)
dev.synchronize()
import gc
gc.collect(); gc.disable()   # a collector pass over the set-up's arrays takes 40–80 ms: inside the loop it doubled the reported time per step (round 5)
t0 = time.perf_counter()
its = 0
per = []
for s in range(args.steps):
    ts = time.perf_counter()
    assert ltg.step((s + 1) * args.dt, args.dt)
    its += heat.last_iters
    if args.per_step:
        dev.synchronize(); per.append((round((time.perf_counter() - ts) * 1e3, 3), heat.last_iters))
if per:
    print("per-step ms, iterations:", per, file=sys.stderr)
dev.synchronize()
el = time.perf_counter() - t0
u = cache.un.to_host()
print(json.dumps({"workload": "monodomain + %s, %d^3 hex Q1 (%d cells, %d dofs), LTG(BE+CG, %s cell solver), dt=%g" % (args.ionic.upper(), args.n, g.n_cells, n, type(cell_solver).__name__, args.dt),
                  "setup_incl_initial_assembly_s": t_setup, "ms_per_time_step": el / args.steps * 1e3, "cg_iterations_per_step": its / args.steps,
                  "dof_updates_per_s": model.nstates * n * args.steps / el, "phi_range": [float(u[:n].min()), float(u[:n].max())]}))
