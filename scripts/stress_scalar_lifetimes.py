#!/usr/bin/env python3
"""Create / use / destroy the scalar-form operators whose plans were added in round 2 — fused hexahedron pair, tetrahedron patch kernel, quadratic-field
element strategy (fixed-width slot table, cell-major coordinates), vector patch plans — in a loop and watch the device memory.  One JSON line."""
import gc, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import thunderbolt_jl_amd as tb
dev = tb.MI355XDevice(0)
T = [(0, 1, 2, 6), (0, 2, 3, 6), (0, 3, 7, 6), (0, 7, 4, 6), (0, 4, 5, 6), (0, 5, 1, 6)]


def free_bytes():
    dev.synchronize()
    return torch.cuda.mem_get_info()[0]


def one_round():
    chk = 0.0
    g = tb.generate_mesh(tb.Hexahedron, (14, 12, 10), (0, 0, 0), (1.0, 0.8, 0.6), perturb=0.1)
    D = np.array([[2.0, 0.3, 0.1], [0.3, 1.5, -0.2], [0.1, -0.2, 1.0]])
    # hexahedra, first order: fused pair + vector patch kernels
    dh = tb.DofHandler(g); sp = tb.allocate_matrix(dh)
    st = tb.PatchAssemblyStrategy(dev)
    M = tb.setup_operator(st, tb.BilinearMassIntegrator(tb.ConstantCoefficient(1.0)), dh, sp)
    K = tb.setup_operator(st, tb.BilinearDiffusionIntegrator(tb.ConstantCoefficient(D)), dh, sp)
    tb.update_operators(M, K, 0.0)
    for s in (tb.PatchAssemblyStrategy(dev), tb.AtomicAssemblyStrategy(dev)):
        b = tb.update_operator(tb.setup_operator(s, tb.LinearIntegrator(tb.AnalyticalCoefficient("cos_exp")), dh), 0.1)
        chk += float(b.b.to_host().sum())
    chk += float(M.A.to_host().sum()) + float(K.A.to_host().sum())
    # quadratic field: element strategy
    dh2 = tb.DofHandler(g, tb.LagrangeCollection(2)); sp2 = tb.allocate_matrix(dh2)
    for s in (tb.ElementAssemblyStrategy(dev), tb.AtomicAssemblyStrategy(dev)):
        M2 = tb.update_operator(tb.setup_operator(s, tb.BilinearMassIntegrator(tb.ConstantCoefficient(1.0)), dh2, sp2), 0.0)
        K2 = tb.update_operator(tb.setup_operator(s, tb.BilinearDiffusionIntegrator(tb.ConstantCoefficient(D)), dh2, sp2), 0.0)
        chk += float(M2.A.to_host().sum()) + float(K2.A.to_host().sum())
    # tetrahedra: staged patch kernel, pair
    tets = np.stack([g.conn[:, list(t)] for t in T], axis=1).reshape(-1, 4).astype(np.int32)
    X = g.xyz[tets]
    flip = np.einsum("ij,ij->i", np.cross(X[:, 1] - X[:, 0], X[:, 2] - X[:, 0]), X[:, 3] - X[:, 0]) < 0
    tets[flip] = tets[flip][:, [0, 2, 1, 3]]
    gt = tb.Grid(tb.Tetrahedron, g.xyz, np.ascontiguousarray(tets))
    dht = tb.DofHandler(gt); spt = tb.allocate_matrix(dht)
    Mt = tb.setup_operator(st, tb.BilinearMassIntegrator(tb.ConstantCoefficient(1.0)), dht, spt)
    Kt = tb.setup_operator(st, tb.BilinearDiffusionIntegrator(tb.ConstantCoefficient(D)), dht, spt)
    tb.update_operators(Mt, Kt, 0.0)
    chk += float(Mt.A.to_host().sum()) + float(Kt.A.to_host().sum())
    return chk


one_round(); gc.collect()
f0 = free_bytes()
ok = True
for r in range(15):
    ok = ok and np.isfinite(one_round())
    gc.collect()
f1 = free_bytes()
print(json.dumps({"rounds": 15, "free_before": f0, "free_after": f1, "leaked_bytes": f0 - f1, "finite": bool(ok)}))
