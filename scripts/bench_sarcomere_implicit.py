#!/usr/bin/env python3
"""Timing of the pointwise backward-Euler sarcomere solve (tb_sarcomere_implicit_step) on one MI355X; prints one JSON line."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=1728000)
ap.add_argument("--reps", type=int, default=5)
args = ap.parse_args()
import thunderbolt_jl_amd as tb
dev = tb.MI355XDevice(0)
n = args.points
rng = np.random.default_rng(0)
model = tb.RDQ20MFModel()
lam, ca = dev.to_device(rng.uniform(0.9, 1.1, n)), dev.to_device(rng.uniform(0.1, 0.9, n))
known = tb.SarcomereState(dev, model, n)
out = {}
for name, ls, sens in (("1_iteration_no_corrector", tb.GenericLocalNonlinearSolver(1, 1e-30), False), ("default_tol_1e-4", None, False),
                       ("default_tol_1e-4_with_corrector", None, True), ("tol_1e-12_with_corrector", tb.GenericLocalNonlinearSolver(30, 1e-12), True)):
    dQ = dev.zeros(20 * n) if sens else None
    ts = []
    for _ in range(args.reps):
        st = tb.SarcomereState(dev, model, n)
        dev.synchronize()
        t0 = time.perf_counter()
        tb.sarcomere_implicit_step(st, known, 0.5, lam, ca, ls, dstate_dstretch=dQ, count_failures=False)
        dev.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    out[name] = min(ts)
print(json.dumps({"workload": "RDQ20-MF backward Euler, %d points, dt 0.5 from the resting state" % n, "ms": out}))
