import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import thunderbolt_jl_amd as tb
dev = tb.MI355XDevice(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
g = tb.generate_mesh(tb.Hexahedron, (n, n, n), (0, 0, 0), (1, 1, 1), perturb=0.2)
dh = tb.DofHandler(g); sp = tb.allocate_matrix(dh)
kap = np.array([[4.5e-5, 1e-5, 0], [1e-5, 2.0e-5, 0], [0, 0, 2.0e-5]])
ops = {name: tb.setup_operator(S(dev), tb.BilinearDiffusionIntegrator(tb.ConstantCoefficient(kap)), dh, sp)
       for name, S in (("patch", tb.PatchAssemblyStrategy), ("atomic", tb.AtomicAssemblyStrategy), ("color", tb.PerColorAssemblyStrategy))}
ref = tb.update_operator(ops["patch"], 0.0).A.to_host()
scale = np.abs(ref).max()
nbad = 0
for rep in range(reps):
    for name, op in ops.items():
        # fresh operator every few reps to exercise lazy plan builds too
        if rep % 5 == 4:
            op = tb.setup_operator(type(op.strategy)(dev), tb.BilinearDiffusionIntegrator(tb.ConstantCoefficient(kap)), dh, sp)
        v = tb.update_operator(op, 0.0).A.to_host()
        d = np.abs(v - ref)
        bad = np.nonzero(d > 1e-10 * scale)[0]
        if len(bad):
            nbad += 1
            rows = np.searchsorted(sp.rowptr, bad, side="right") - 1
            print("MISMATCH rep", rep, name, "bad entries", len(bad), "first", bad[:8], "rows", rows[:8], "ratio", (v[bad[:8]] / ref[bad[:8]]))
print("done; mismatching runs:", nbad, "of", reps * 3)
