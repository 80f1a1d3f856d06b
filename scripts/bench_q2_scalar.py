#!/usr/bin/env python3
"""Scalar Q2 forms (27-dof hexahedra) on one MI355X: mass, diffusion, source on an n³ mesh; prints one JSON line."""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=64)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--strategies", default="atomic,color,element")
args = ap.parse_args()
import thunderbolt_jl_amd as tb
dev = tb.MI355XDevice(0)
g = tb.generate_mesh(tb.Hexahedron, (args.n,) * 3, (0, 0, 0), (1.0, 1.0, 1.0), perturb=0.1)
dh = tb.DofHandler(g, tb.LagrangeCollection(2))
sp = tb.allocate_matrix(dh)
out = {"workload": "scalar Q2, %d^3 hex (%d cells, %d dofs, nnz %d)" % (args.n, g.n_cells, dh.ndofs, sp.nnz)}
D = np.array([[2.0, 0.3, 0.1], [0.3, 1.5, -0.2], [0.1, -0.2, 1.0]])
for sname, st in (("atomic", tb.AtomicAssemblyStrategy(dev)), ("color", tb.PerColorAssemblyStrategy(dev)), ("element", tb.ElementAssemblyStrategy(dev))):
    if sname not in args.strategies.split(","):
        continue
    ops = {"mass": tb.setup_operator(st, tb.BilinearMassIntegrator(tb.ConstantCoefficient(1.0)), dh, sp),
           "diffusion": tb.setup_operator(st, tb.BilinearDiffusionIntegrator(tb.ConstantCoefficient(D)), dh, sp),
           "source": tb.setup_operator(st, tb.LinearIntegrator(tb.AnalyticalCoefficient("cos_exp")), dh)}
    for name, op in ops.items():
        tb.update_operator(op, 0.1)
        from _preroll import preroll
        preroll(dev, lambda: tb.update_operator(op, 0.1), 100.0)   # steady clocks
        e0, e1 = dev.event(), dev.event()
        e0.record()
        for _ in range(args.steps):
            tb.update_operator(op, 0.1)
        e1.record()
        ms = e0.elapsed_ms(e1) / args.steps
        out["%s_%s" % (name, sname)] = {"ms": ms, "cells_per_s": g.n_cells / ms * 1e3}
print(json.dumps(out))
