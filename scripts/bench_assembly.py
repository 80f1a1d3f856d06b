#!/usr/bin/env python3
"""Times the Q1 matrix assembly paths at one size: fused M+K pass, the same kernel per matrix, and (TB_PATCH_KERNEL=legacy) the general
patch kernel.  Tile shape via TB_PATCH_TILE=tx,ty,tz.  usage: python3 scripts/bench_assembly.py [--n 216] [--reps 10]"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import thunderbolt_jl_amd as tb  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=216)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--check", action="store_true", help="compare fused vs separate results")
ap.add_argument("--fibre", action="store_true", help="diffusion with a nodal fibre field (SpectralTensorCoefficient over OrthotropicMicrostructureModel)")
a = ap.parse_args()
dev = tb.MI355XDevice(0)
g = tb.generate_mesh(tb.Hexahedron, (a.n,) * 3, (0, 0, 0), (1, 1, 1), perturb=0.2)
dh = tb.DofHandler(g)
sp = tb.allocate_matrix(dh)
st = tb.PatchAssemblyStrategy(dev)
kap = np.diag([4.5e-5, 2.0e-5, 2.0e-5])
D = tb.ConductivityToDiffusivityCoefficient(tb.ConstantCoefficient(kap), tb.ConstantCoefficient(1.0), tb.ConstantCoefficient(1.0))
if a.fibre:
    rng = np.random.default_rng(0)
    nc = g.n_cells
    f = rng.normal(size=(nc, 8, 3)) * 0.2 + np.array([2.0, 0, 0]); s_ = rng.normal(size=(nc, 8, 3)) * 0.2 + np.array([0, 2.0, 0]); n_ = rng.normal(size=(nc, 8, 3)) * 0.2 + np.array([0, 0, 2.0])
    D = tb.ConductivityToDiffusivityCoefficient(tb.SpectralTensorCoefficient(tb.OrthotropicMicrostructureModel(f, s_, n_), tb.ConstantCoefficient(np.array([4.5e-5, 2.0e-5, 1.0e-5]))),
                                                tb.ConstantCoefficient(1.0), tb.ConstantCoefficient(1.0))
M = tb.setup_operator(st, tb.BilinearMassIntegrator(tb.ConstantCoefficient(1.0)), dh, sp)
K = tb.setup_operator(st, tb.BilinearDiffusionIntegrator(D), dh, sp)


def timeit(fn):
    fn(); fn()
    e0, e1 = dev.event(), dev.event()
    e0.record()
    for _ in range(a.reps):
        fn()
    e1.record()
    dev.synchronize()
    return e0.elapsed_ms(e1) / a.reps


out = {"n": a.n, "tile": os.environ.get("TB_PATCH_TILE", "default"), "kernel": os.environ.get("TB_PATCH_KERNEL", "new")}
order = os.environ.get("TB_BENCH_ORDER", "sep,fused").split(",")
for what in order:
    if what == "sep":
        out["mass_ms"] = timeit(lambda: tb.update_operator(M, 0.0))
        out["diffusion_ms"] = timeit(lambda: tb.update_operator(K, 0.0))
        if a.check:
            Mh, Kh = M.A.to_host(), K.A.to_host()
    elif what == "fused":
        out["fused_ms"] = timeit(lambda: tb.update_operators(M, K, 0.0))
        if a.check and "mass_ms" in out:
            out["fused_vs_sep"] = [float(np.abs(M.A.to_host() - Mh).max() / np.abs(Mh).max()), float(np.abs(K.A.to_host() - Kh).max() / np.abs(Kh).max())]
if "vec" in order:
    for name, st_ in (("atomic", tb.AtomicAssemblyStrategy(dev)), ("patch", tb.PatchAssemblyStrategy(dev))):
        src = tb.setup_operator(st_, tb.LinearIntegrator(tb.AnalyticalCoefficient("cos_exp")), dh)
        out["source_%s_ms" % name] = timeit(lambda: tb.update_operator(src, 0.1))
print(json.dumps(out))
