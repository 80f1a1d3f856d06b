#!/usr/bin/env python3
"""Chunked mechanics linearisation under repetition: the integration launches on the device's stream and the gathers on the second queue must give
the same tangent whatever the number of chunks and whatever runs right before and after (a product with the tangent on the main stream, the next
linearisation).  40³ Q2 hexahedra; the product J·t of every round is compared bit for bit with the first one.  Prints one JSON line."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import thunderbolt_jl_amd as tb
dev = tb.MI355XDevice(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 120
g = tb.generate_mesh(tb.Hexahedron, (n, n, n), (0, 0, 0), (1.0, 1.0, 1.0), perturb=0.1)
dh = tb.DofHandler(g, tb.LagrangeCollection(2) ** 3)
sp = tb.allocate_matrix(dh)
f, s_, nn = np.array([1, 1, 0.0]) / np.sqrt(2), np.array([-1, 1, 0.0]) / np.sqrt(2), np.array([0, 0, 1.0])
model = tb.QuasiStaticModel("u", tb.PK1Model(tb.HolzapfelOgden2009Model(), tb.ConstantCoefficient(tb.OrthotropicMicrostructure(f, s_, nn))))
op = tb.setup_operator(tb.ElementAssemblyStrategy(dev), model, dh, sp)
u = dev.to_device(1e-2 * np.sin(np.pi * np.arange(dh.ndofs) / dh.ndofs))
t = dev.to_device(np.cos(1e-3 * np.arange(dh.ndofs)))
res, y = dev.zeros(dh.ndofs), dev.zeros(dh.ndofs)
L = tb.lib()
os.environ["TB_MECH_CHUNKS"] = "0"
tb.update_linearization(op, u, 0.0, residual=res)
tb.check(L.tb_spmv_csr(op.pattern.h, op.J.ptr, t.ptr, 1.0, 0.0, y.ptr))
y0, r0 = y.to_host(), res.to_host()
bad = 0
for k in range(rounds):
    os.environ["TB_MECH_CHUNKS"] = str((0, 2, 3, 5, 8, 16)[k % 6])
    tb.update_linearization(op, u, 0.0, residual=res)
    tb.check(L.tb_spmv_csr(op.pattern.h, op.J.ptr, t.ptr, 1.0, 0.0, y.ptr))     # reads the tangent on the main stream right behind the gathers
    if k % 3 == 0:
        tb.update_linearization(op, u, 0.0, residual=res)                       # … and a linearisation right behind a linearisation
        tb.check(L.tb_spmv_csr(op.pattern.h, op.J.ptr, t.ptr, 1.0, 0.0, y.ptr))
    bad += int(not np.array_equal(y.to_host(), y0)) + int(not np.array_equal(res.to_host(), r0))
print(json.dumps({"cells": g.n_cells, "dofs": dh.ndofs, "rounds": rounds, "mismatches": bad}))
sys.exit(1 if bad else 0)
