#!/usr/bin/env python3
"""The reference's first tutorial on one MI355X (docs/src/literate-tutorials/ep01_spiral-wave.jl, line by line):
generate_mesh(Quadrilateral, (2⁶, 2⁶), (0,0), (2.5,2.5)); monodomain with Cₘ = χ = 1, κ = diag(4.5e-5, 2e-5), FitzHugh–Nagumo;
ReactionDiffusionSplit; φₘ = 1 on x, y ≤ 1.25, s = 0.1 on y ≥ 1.25; LieTrotterGodunov((BackwardEulerSolver(CG atol 1e-6,
rtol 1e-5), AdaptiveForwardEulerSubstepper(reaction_threshold = 0.1))), Δt = 1, tspan (0, 1000).  Prints one JSON line."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=2 ** 6)
ap.add_argument("--tend", type=float, default=1000.0)
args = ap.parse_args()
import thunderbolt_jl_amd as tb
dev = tb.MI355XDevice(0)
mesh = tb.generate_mesh(tb.Quadrilateral, (args.n, args.n), (0.0, 0.0), (2.5, 2.5))
dh = tb.DofHandler(mesh)
kappa = tb.ConductivityToDiffusivityCoefficient(tb.ConstantCoefficient(np.array([[4.5e-5, 0.0], [0.0, 2.0e-5]])), tb.ConstantCoefficient(1.0),
                                                tb.ConstantCoefficient(1.0))
heat = tb.BackwardEulerStage(tb.BackwardEulerSolver(atol=1e-6, rtol=1e-5), tb.PatchAssemblyStrategy(dev), dh, kappa)
model = tb.FHNModel()
n = dh.ndofs
X = np.empty((n, 3)); X[tb.distributed.node_to_dof(dh)] = mesh.xyz
u0 = np.zeros((2, n))
u0[0] = ((X[:, 0] <= 1.25) & (X[:, 1] <= 1.25)).astype(float)
u0[1] = 0.1 * (X[:, 1] >= 1.25)
f = tb.PointwiseODEFunction(n, model)
cache = tb.setup_solver_cache(f, tb.AdaptiveForwardEulerSubstepper(dev, reaction_threshold=0.1), u=dev.to_device(u0.ravel()), keep_du=False)
ltg = tb.LieTrotterGodunov(heat, f, cache)
dt, t, steps, its = 1.0, 0.0, 0, 0
t0 = time.perf_counter()
while t < args.tend - 1e-9:
    assert ltg.step(t, dt)
    its += heat.last_iters
    t += dt; steps += 1
dev.synchronize()
el = time.perf_counter() - t0
phi = cache.un.to_host()[:n]
print(json.dumps({"workload": "ep01 spiral wave, %d² quadrilaterals, FHN, LTG(BE+CG, adaptive FE), dt=1, t_end=%g" % (args.n, args.tend),
                  "time_steps": steps, "ms_per_time_step": el / steps * 1e3, "cg_iterations_per_step": its / steps,
                  "phi_range": [float(phi.min()), float(phi.max())], "activated_fraction": float((phi > 0.5).mean())}))
