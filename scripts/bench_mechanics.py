#!/usr/bin/env python3
"""BASELINE config 4: quasi-static Holzapfel–Ogden mechanics, Q2 displacement on an n³ hex mesh, Newton
residual / tangent assembly on one MI355X.  Prints one JSON line (element-integrations/s for
update_linearization! and residual!)."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=40)
ap.add_argument("--order", type=int, default=2)
ap.add_argument("--hill", default="", choices=["", "extended", "generalized"], help="ExtendedHillModel(HO2009, LinearSpring, GMK, PSL1995) / GeneralizedHillModel(LinYin passive + active, GMK incompressible)")
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--strategy", default="element", choices=["atomic", "color", "element"])
ap.add_argument("--cpu-n", type=int, default=8)
ap.add_argument("--cell-order", default="lexicographic", choices=["lexicographic", "morton"], help="order of the cells in memory (element matrices of the element strategy are stored in cell order)")
ap.add_argument("--spmv", action="store_true", help="also time the product J·x with the assembled tangent (the block SpMV of the Krylov solvers)")
ap.add_argument("--energy", default="ho", choices=["ho", "guccione", "humphrey", "linyin"],
                help="passive energy: ho = Holzapfel–Ogden 2009 with hand-derived P, 𝔸 (fast path); the others run the generic instance (hyper-dual differentiation of Ψ per pair of components of F)")
ap.add_argument("--chunks", default="", help="comma list: also time the linearisation with TB_MECH_CHUNKS set to each value (gather of chunk k beside the integration of chunk k + 1)")
ap.add_argument("--condensed", action="store_true", help="active stress with the RDQ20-MF internal state condensed per quadrature point")
args = ap.parse_args()
import thunderbolt_jl_amd as tb
dev = tb.MI355XDevice(0)
t0 = time.time()
g = tb.generate_mesh(tb.Hexahedron, (args.n,) * 3, (0, 0, 0), (1, 1, 1), perturb=0.1)
if args.cell_order == "morton":
    def spread(v):
        v = v.astype(np.uint64) & np.uint64(0x1fffff)
        v = (v | v << np.uint64(32)) & np.uint64(0x1f00000000ffff)
        v = (v | v << np.uint64(16)) & np.uint64(0x1f0000ff0000ff)
        v = (v | v << np.uint64(8)) & np.uint64(0x100f00f00f00f00f)
        v = (v | v << np.uint64(4)) & np.uint64(0x10c30c30c30c30c3)
        return (v | v << np.uint64(2)) & np.uint64(0x1249249249249249)
    c = np.arange(g.n_cells)
    key = spread(c % args.n) | spread((c // args.n) % args.n) << np.uint64(1) | spread(c // (args.n * args.n)) << np.uint64(2)
    g = tb.Grid(tb.Hexahedron, g.xyz, np.ascontiguousarray(g.conn[np.argsort(key, kind="stable")]))
dh = tb.DofHandler(g, tb.LagrangeCollection(args.order) ** 3)
sp = tb.allocate_matrix(dh)
t_setup = time.time() - t0
energy = {"ho": tb.HolzapfelOgden2009Model, "guccione": tb.Guccione1991PassiveModel, "humphrey": tb.HumphreyStrumpfYinModel, "linyin": tb.LinYinPassiveModel}[args.energy]()
model = tb.QuasiStaticModel("u", tb.PK1Model(energy, tb.ConstantCoefficient(tb.OrthotropicMicrostructure([1, 0, 0], [0, 1, 0], [0, 0, 1]))))
if args.hill:
    ms_ = tb.ConstantCoefficient(tb.OrthotropicMicrostructure([1, 0, 0], [0, 1, 0], [0, 0, 1]))
    sarc = tb.CaDrivenInternalSarcomereModel(tb.PelceSunLangeveld1995Model(), 0.6)
    cm = (tb.ExtendedHillModel(tb.HolzapfelOgden2009Model(), tb.ActiveMaterialAdapter(tb.LinearSpringModel()), tb.GMKActiveDeformationGradientModel(), sarc, ms_)
          if args.hill == "extended" else
          tb.GeneralizedHillModel(tb.LinYinPassiveModel(), tb.ActiveMaterialAdapter(tb.LinYinActiveModel()), tb.GMKIncompressibleActiveDeformationGradientModel(), sarc, ms_))
    model = tb.QuasiStaticModel("u", cm)
if args.condensed:
    fsn = tb.ConstantCoefficient(tb.OrthotropicMicrostructure([1, 0, 0], [0, 1, 0], [0, 0, 1]))
    model = tb.QuasiStaticModel("u", tb.ActiveStressModel(tb.HolzapfelOgden2009Model(), tb.SimpleActiveStress(Tmax=50.0),
                                                      tb.CaDrivenInternalSarcomereModel(tb.AsRateIndependent(tb.RDQ20MFModel()), 0.6), fsn))
st = {"atomic": tb.AtomicAssemblyStrategy, "color": tb.PerColorAssemblyStrategy, "element": tb.ElementAssemblyStrategy}[args.strategy](dev)
op = tb.setup_operator(st, model, dh, sp)
if args.condensed:
    tb.set_timestep(op, 0.5)
u = 1e-2 * np.sin(np.pi * np.arange(dh.ndofs) / dh.ndofs)   # smooth small displacement state
du, res = dev.to_device(u), dev.zeros(dh.ndofs)
tb.update_linearization(op, du, 0.0, residual=res)          # warm-up (builds block positions / colours)
tb.residual(op, res, du, 0.0)
from _preroll import preroll
preroll(dev, lambda: (tb.update_linearization(op, du, 0.0, residual=res), tb.residual(op, res, du, 0.0)))   # steady clocks
e = [dev.event() for _ in range(3)]
tl = tr = 0.0
for _ in range(args.steps):
    e[0].record(); tb.update_linearization(op, du, 0.0, residual=res)
    e[1].record(); tb.residual(op, res, du, 0.0)
    e[2].record()
    tl += e[0].elapsed_ms(e[1]); tr += e[1].elapsed_ms(e[2])
tl /= args.steps; tr /= args.steps
out = {"workload": ("HO2009 + condensed RDQ20-MF active stress, " if args.condensed else "") + ("HO2009" if args.energy == "ho" else args.energy + " (generic hyper-dual instance)") + " quasi-static, Q%d displacement, %d^3 hex (%d cells, %d dofs, nnz %d), %s scatter" % (args.order, args.n, g.n_cells, dh.ndofs, sp.nnz, args.strategy),
       "linearize_ms": tl, "residual_ms": tr, "linearize_cells_per_s": g.n_cells / (tl * 1e-3), "residual_cells_per_s": g.n_cells / (tr * 1e-3),
       "host_setup_s": t_setup, "cell_order": args.cell_order}
if args.chunks:
    import os
    out["linearize_ms_by_chunks"] = {}
    for c in args.chunks.split(","):
        os.environ["TB_MECH_CHUNKS"] = c
        tb.update_linearization(op, du, 0.0, residual=res)
        e[0].record()
        for _ in range(args.steps):
            tb.update_linearization(op, du, 0.0, residual=res)
        e[1].record(); dev.synchronize()
        out["linearize_ms_by_chunks"][c] = e[0].elapsed_ms(e[1]) / args.steps
    os.environ.pop("TB_MECH_CHUNKS")
if args.spmv:
    xs, ys = dev.to_device(np.cos(np.arange(dh.ndofs) * 1e-3)), dev.zeros(dh.ndofs)
    def prod():
        tb._lib.check(tb.lib().tb_spmv_csr(op.pattern.h, op.J.ptr, xs.ptr, 1.0, 0.0, ys.ptr))
    prod(); prod()
    ea, eb = dev.event(), dev.event()
    ea.record()
    for _ in range(10):
        prod()
    eb.record(); dev.synchronize()
    ms = ea.elapsed_ms(eb) / 10
    out["spmv_ms"] = ms
    out["spmv_TBps_8.4B_per_nz"] = (8.0 + 4.0 / 9.0) * sp.nnz / (ms * 1e-3) / 1e12
if args.condensed or args.energy != "ho":
    if args.condensed:
        out["quadrature_points"] = op.internal.n_points
    print(json.dumps(out)); sys.exit(0)
# CPU oracle ("port", C restatement with hyper-dual AD — not Julia) on a bounded sample
from oracle import oracle as o
n = args.cpu_n
xyz, conn = o.generate_grid_hex(n, n, n, (0, 0, 0), (1, 1, 1))
kind, q = (o.HEX27, 3) if args.order == 2 else (o.HEX8, 2)
cd, nd = o.close_dofs(kind, 3, conn, len(xyz))
rp, ci = o.build_pattern(cd, nd)
col, nc = o.color_cells(cd, nd)
m = o.Mesh(kind, q, xyz, conn, cd)
uu = 1e-2 * np.sin(np.pi * np.arange(nd) / nd)
th = min(32, len(os.sched_getaffinity(0)))
t0 = time.perf_counter(); o.assemble_hyperelastic(m, uu, rp, ci, nthreads=th, color=col, ncolors=nc); t1 = time.perf_counter()
out["cpu_baseline"] = {"value": n ** 3 / (t1 - t0), "unit": "element-integrations/s (K+r)", "cores": th, "kind": "port", "sample": "%d^3 cells" % n}
print(json.dumps(out))
