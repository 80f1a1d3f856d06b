#!/usr/bin/env python3
"""Linear tetrahedra (the other cell type of the north star's unstructured hex/tet meshes) on one MI355X: mass, diffusion, source on an n³ hexahedral grid
cut into 6 n³ tetrahedra, cells ordered hex by hex (the six tetrahedra of a hexahedron are neighbours in memory); every strategy; one JSON line."""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=100)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--strategies", default="patch,atomic,color,element")
args = ap.parse_args()
import thunderbolt_jl_amd as tb
dev = tb.MI355XDevice(0)
g0 = tb.generate_mesh(tb.Hexahedron, (args.n,) * 3, (0, 0, 0), (1.0, 1.0, 1.0), perturb=0.1)
T = [(0, 1, 2, 6), (0, 2, 3, 6), (0, 3, 7, 6), (0, 7, 4, 6), (0, 4, 5, 6), (0, 5, 1, 6)]
tets = np.stack([g0.conn[:, list(t)] for t in T], axis=1).reshape(-1, 4).astype(np.int32)
X = g0.xyz[tets]
flip = np.einsum("ij,ij->i", np.cross(X[:, 1] - X[:, 0], X[:, 2] - X[:, 0]), X[:, 3] - X[:, 0]) < 0
tets[flip] = tets[flip][:, [0, 2, 1, 3]]
g = tb.Grid(tb.Tetrahedron, g0.xyz, np.ascontiguousarray(tets))
dh = tb.DofHandler(g)
sp = tb.allocate_matrix(dh)
out = {"workload": "scalar P1 on tetrahedra, %d^3 x 6 (%d cells, %d dofs, nnz %d)" % (args.n, g.n_cells, dh.ndofs, sp.nnz),
       "algorithmic_bytes_per_cell": (16 * g.n_cells + 24 * dh.ndofs + 8 * sp.nnz) / g.n_cells}
D = np.array([[2.0, 0.3, 0.1], [0.3, 1.5, -0.2], [0.1, -0.2, 1.0]])
S = {"patch": tb.PatchAssemblyStrategy, "atomic": tb.AtomicAssemblyStrategy, "color": tb.PerColorAssemblyStrategy, "element": tb.ElementAssemblyStrategy}
for sname in args.strategies.split(","):
    st = S[sname](dev)
    ops = {"mass": tb.setup_operator(st, tb.BilinearMassIntegrator(tb.ConstantCoefficient(1.0)), dh, sp),
           "diffusion": tb.setup_operator(st, tb.BilinearDiffusionIntegrator(tb.ConstantCoefficient(D)), dh, sp),
           "source": tb.setup_operator(st, tb.LinearIntegrator(tb.AnalyticalCoefficient("cos_exp")), dh)}
    for name, op in ops.items():
        try:
            tb.update_operator(op, 0.1)
        except tb.TBError as e:
            out["%s_%s" % (name, sname)] = "unsupported: %s" % str(e)[:80]
            continue
        from _preroll import preroll
        preroll(dev, lambda: tb.update_operator(op, 0.1), 60.0)   # steady clocks
        e0, e1 = dev.event(), dev.event()
        e0.record()
        for _ in range(args.steps):
            tb.update_operator(op, 0.1)
        e1.record()
        ms = e0.elapsed_ms(e1) / args.steps
        out["%s_%s" % (name, sname)] = {"ms": ms, "cells_per_s": g.n_cells / ms * 1e3}
print(json.dumps(out))
