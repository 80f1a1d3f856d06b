#!/usr/bin/env python3
"""Create / use / destroy the mechanics operators (plain, Hill, prestressed, subdomains, condensed rate-free and rate-coupled) in a loop
and watch the device memory: a leak or a use-after-free in the form / pattern lifetimes shows up here.  Prints one JSON line."""
import gc, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import thunderbolt_jl_amd as tb
dev = tb.MI355XDevice(0)
rng = np.random.default_rng(0)


def free_bytes():
    dev.synchronize()
    return torch.cuda.mem_get_info()[0]


def one_round(order):
    g = tb.generate_mesh(tb.Hexahedron, (6, 5, 4), (0, 0, 0), (1.0, 0.8, 0.6), perturb=0.1)
    g.addcellset("a", np.arange(0, g.n_cells, 2)); g.addcellset("b", np.arange(1, g.n_cells, 2))
    dh = tb.DofHandler(g, tb.LagrangeCollection(order) ** 3)
    sp = tb.allocate_matrix(dh)
    ms = tb.ConstantCoefficient(tb.OrthotropicMicrostructure([1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0]))
    ho, gu = tb.HolzapfelOgden2009Model, tb.Guccione1991PassiveModel
    sarc = tb.CaDrivenInternalSarcomereModel(tb.PelceSunLangeveld1995Model(), 0.5)
    models = [
        (tb.ElementAssemblyStrategy, tb.QuasiStaticModel("u", tb.PK1Model(ho(), ms), [tb.RobinBC(0.1, "right"), tb.ConstantPressureBC(0.01, "top")])),
        (tb.PerColorAssemblyStrategy, tb.QuasiStaticModel("u", tb.ExtendedHillModel(ho(), tb.ActiveMaterialAdapter(tb.LinearSpringModel()), tb.GMKActiveDeformationGradientModel(), sarc, ms))),
        (tb.ElementAssemblyStrategy, tb.QuasiStaticModel("u", tb.PrestressedMechanicalModel(tb.PK1Model(gu(), ms), tb.ConstantCoefficient(np.eye(3) * 1.05)))),
        (tb.AtomicAssemblyStrategy, {"a": tb.QuasiStaticModel("u", tb.PK1Model(ho(), ms)), "b": tb.QuasiStaticModel("u", tb.PK1Model(gu(), ms))}),
        (tb.ElementAssemblyStrategy, tb.QuasiStaticModel("u", tb.ActiveStressModel(ho(), tb.SimpleActiveStress(Tmax=40.0), tb.CaDrivenInternalSarcomereModel(tb.AsRateIndependent(tb.RDQ20MFModel()), 0.6), ms))),
        (tb.PerColorAssemblyStrategy, tb.QuasiStaticModel("u", tb.ActiveStressModel(gu(), tb.SimpleActiveStress(Tmax=40.0), tb.CaDrivenInternalSarcomereModel(tb.RDQ20MFModel(), 0.6), ms))),
    ]
    u = dev.to_device(rng.uniform(-1e-2, 1e-2, dh.ndofs))
    chk = 0.0
    for st, model in models:
        op = tb.setup_operator(st(dev), model, dh, sp)
        if op.internal is not None:
            tb.set_timestep(op, 0.5)
        r = dev.zeros(dh.ndofs)
        for _ in range(3):
            tb.update_linearization(op, u, 0.0, residual=r)
            tb.residual(op, r, u, 0.0)
        chk += float(np.abs(r.to_host()).sum()) + float(np.abs(op.J.to_host()).sum())
        del op, r
    del u
    gc.collect()
    return chk


one_round(1); one_round(2)
base = free_bytes()
chks = []
for it in range(20):
    chks.append(one_round(1 + it % 2))
end = free_bytes()
print(json.dumps({"rounds": 20, "free_before": base, "free_after": end, "leaked_bytes": base - end, "finite": bool(np.isfinite(chks).all())}))
