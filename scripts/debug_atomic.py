import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import thunderbolt_jl_amd as tb
dev = tb.MI355XDevice(0)
for n in (64, 100):
    g = tb.generate_mesh(tb.Hexahedron, (n, n, n), (0, 0, 0), (1, 1, 1), perturb=0.2)
    dh = tb.DofHandler(g); sp = tb.allocate_matrix(dh)
    kap = np.array([[4.5e-5, 1e-5, 0], [1e-5, 2.0e-5, 0], [0, 0, 2.0e-5]])
    res = {}
    for name, S in (("patch", tb.PatchAssemblyStrategy), ("atomic", tb.AtomicAssemblyStrategy), ("color", tb.PerColorAssemblyStrategy)):
        op = tb.setup_operator(S(dev), tb.BilinearDiffusionIntegrator(tb.ConstantCoefficient(kap)), dh, sp)
        for rep in range(2):
            tb.update_operator(op, 0.0)
            res[(name, rep)] = op.A.to_host()
    ref = res[("patch", 0)]
    for k, v in res.items():
        d = np.abs(v - ref)
        bad = np.nonzero(d > 1e-10 * np.abs(ref).max())[0]
        print(n, k, "max rel", d.max() / np.abs(ref).max(), "bad entries", len(bad), bad[:5], (v[bad[:5]] / ref[bad[:5]]) if len(bad) else "")
