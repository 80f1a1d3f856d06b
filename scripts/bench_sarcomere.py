#!/usr/bin/env python3
"""RDQ20-MF sarcomere step on one MI355X: n points, per-point inputs, forward Euler; prints one JSON line."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=16 * 1024 * 1024)
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--substeps", type=int, default=1)
args = ap.parse_args()
import thunderbolt_jl_amd as tb
dev = tb.MI355XDevice(0)
n = args.points
rng = np.random.default_rng(0)
model = tb.RDQ20MFModel()
st = tb.SarcomereState(dev, model, n)
lam, vel, ca = dev.to_device(rng.uniform(0.9, 1.1, n)), dev.to_device(rng.normal(size=n) * 1e-3), dev.to_device(rng.uniform(0.1, 0.9, n))
Ta, As = dev.zeros(n), dev.zeros(n)
for _ in range(5):
    tb.sarcomere_step(st, 0.0, 1e-3, lam, vel, ca, substeps=args.substeps, tension=Ta, stiffness=As)
dev.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    tb.sarcomere_step(st, 0.0, 1e-3, lam, vel, ca, substeps=args.substeps, tension=Ta, stiffness=As)
dev.synchronize()
ms = (time.perf_counter() - t0) / args.steps * 1e3
bytes_per_point = 2 * 20 * 8 + 3 * 8 + 2 * 8
print(json.dumps({"workload": "RDQ20-MF forward Euler, %d points, %d substeps" % (n, args.substeps), "ms_per_step": ms,
                  "dof_updates_per_s": n * 20 * args.substeps / ms * 1e3, "algorithmic_GBps": n * bytes_per_point / ms / 1e6,
                  "frac_of_8TBps": n * bytes_per_point / ms / 1e6 / 8000.0}))
