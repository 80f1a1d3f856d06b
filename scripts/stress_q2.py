"""Repeat the quadratic-field matrix assemblies and compare every run with the first: the element strategy is bit-reproducible (exact equality),
the scattering strategies agree to rounding.  Looks for races in the persistent, LDS-barrier-only integration kernel (tb_assembly.hip, k_matrix_q2)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import thunderbolt_jl_amd as tb
dev = tb.MI355XDevice(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
g = tb.generate_mesh(tb.Hexahedron, (n, n, n), (0, 0, 0), (1, 1, 1), perturb=0.2)
dh = tb.DofHandler(g, tb.LagrangeCollection(2)); sp = tb.allocate_matrix(dh)
rng = np.random.default_rng(1)
D = np.array([[2.0, 0.3, 0.1], [0.3, 1.5, -0.2], [0.1, -0.2, 1.0]])
N = np.array([[2.0, 0.5, 0.0], [-0.1, 1.5, 0.2], [0.3, 0.0, 1.0]])
kf = rng.uniform(0.2, 3.0, size=(g.n_cells, 8))
forms = {"mass": tb.BilinearMassIntegrator(tb.ConstantCoefficient(1.0)), "mass_field": tb.BilinearMassIntegrator(tb.FieldCoefficient(kf)),
         "diff": tb.BilinearDiffusionIntegrator(tb.ConstantCoefficient(D)), "diff_nonsym": tb.BilinearDiffusionIntegrator(tb.ConstantCoefficient(N)),
         "diff_field": tb.BilinearDiffusionIntegrator(tb.FieldCoefficient(kf))}
nbad = 0
for fname, integ in forms.items():
    ops = {"element": tb.setup_operator(tb.ElementAssemblyStrategy(dev), integ, dh, sp), "atomic": tb.setup_operator(tb.AtomicAssemblyStrategy(dev), integ, dh, sp),
           "color": tb.setup_operator(tb.PerColorAssemblyStrategy(dev), integ, dh, sp)}
    ref = tb.update_operator(ops["element"], 0.0).A.to_host()
    scale = np.abs(ref).max()
    for rep in range(reps):
        for sname, op in ops.items():
            v = tb.update_operator(op, 0.0).A.to_host()
            ok = np.array_equal(v, ref) if sname == "element" else np.abs(v - ref).max() < 1e-12 * scale
            if not ok:
                nbad += 1
                print("MISMATCH", fname, sname, "rep", rep, "max diff", np.abs(v - ref).max() / scale)
print("done; mismatching runs:", nbad, "of", len(forms) * reps * 3)
