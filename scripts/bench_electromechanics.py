#!/usr/bin/env python3
"""BASELINE config 5 on one MI355X at a size that means something: the idealised left ventricle (all-hexahedral, O-grid apex; unstructured for the
patch planner) with ≥ 10⁵ cells, one coupled step = electrophysiology (fibre-aligned monodomain M + K assembly on the nodal fibre field, TT06 reaction
step, one backward-Euler heat solve by the device CG) followed by the mechanics linearisation on the same mesh (ActiveStressModel(Holzapfel–Ogden 2009,
SimpleActiveStress, Ca-driven sarcomere) with the calcium of the EP state, residual + tangent).  No reference counterpart exists, so the
two halves are reported as two kernel rates, as SURVEY §8(d) prescribes for this configuration; parity of the same set-up against the oracle is
tests/test_gpu_parity.py::test_config5_lv_coupled_step.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build(tb, dev, nc, nr, nl, order=1, strategy="patch"):
    g = tb.generate_ideal_lv_mesh_hex(nc, nr, nl)
    f, s, n = tb.ideal_lv_microstructure(g, np.deg2rad(60.0), np.deg2rad(-60.0))
    dhs = tb.DofHandler(g)
    sps = tb.allocate_matrix(dhs)
    lam = np.array([0.3, 0.12, 0.12]) * 1e-2
    D = tb.ConductivityToDiffusivityCoefficient(tb.SpectralTensorCoefficient(tb.OrthotropicMicrostructureModel(f, s, n), tb.ConstantCoefficient(lam)),
                                                tb.ConstantCoefficient(1.0), tb.ConstantCoefficient(1.0))
    st = {"patch": tb.PatchAssemblyStrategy, "color": tb.PerColorAssemblyStrategy, "atomic": tb.AtomicAssemblyStrategy}[strategy](dev)
    M = tb.setup_operator(st, tb.BilinearMassIntegrator(tb.ConstantCoefficient(1.0)), dhs, sps)
    K = tb.setup_operator(st, tb.BilinearDiffusionIntegrator(D), dhs, sps)
    dhv = tb.DofHandler(g, tb.LagrangeCollection(order) ** 3)
    spv = tb.allocate_matrix(dhv)
    calcium = {"field": np.zeros((g.n_cells, 8))}
    sarc = tb.CaDrivenInternalSarcomereModel(tb.PelceSunLangeveld1995Model(), lambda t: calcium["field"])
    cm = tb.ActiveStressModel(tb.HolzapfelOgden2009Model(), tb.SimpleActiveStress(Tmax=20.0), sarc, tb.OrthotropicMicrostructureModel(f, s, n))
    op = tb.setup_operator(tb.ElementAssemblyStrategy(dev), tb.QuasiStaticModel("d", cm), dhv, spv)
    return dict(g=g, fsn=(f, s, n), dhs=dhs, sps=sps, M=M, K=K, D=D, lam=lam, dhv=dhv, spv=spv, op=op, calcium=calcium)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nc", type=int, default=128); ap.add_argument("--nr", type=int, default=8); ap.add_argument("--nl", type=int, default=100)
    ap.add_argument("--order", type=int, default=1, help="displacement order (1 or 2)")
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    import thunderbolt_jl_amd as tb
    dev = tb.MI355XDevice(0)
    t0 = time.perf_counter()
    P = build(tb, dev, args.nc, args.nr, args.nl, args.order)
    g, dhs, dhv, M, K, op = P["g"], P["dhs"], P["dhv"], P["M"], P["K"], P["op"]
    model = tb.TT06()
    npts = dhs.ndofs
    n2d = tb.distributed.node_to_dof(dhs)
    u0 = np.tile(model.default_initial_state(), (npts, 1)).T.copy()
    X = np.empty((npts, 3)); X[n2d] = g.xyz
    apex = g.xyz[g.getnodeset("Apex")[0]]
    u0[model.phi_index, np.linalg.norm(X - apex, axis=1) < 0.45] = 20.0
    f = tb.PointwiseODEFunction(npts, model)
    cache = tb.setup_solver_cache(f, tb.RushLarsenCellSolver(dev), u=dev.to_device(np.ascontiguousarray(u0).ravel()), keep_du=False)
    setup_s = time.perf_counter() - t0
    dt = 0.05
    ev = [dev.event() for _ in range(6)]
    # warm-up (plans, tables), with the status read after every call
    tb.update_operators(M, K, 0.0)
    tb.perform_step(f, cache, 0.0, dt)
    A = tb.heat_system_matrix(dev, M, K, dt)
    ud = dev.zeros(dhv.ndofs)
    rng = np.random.default_rng(0)
    ud.copy_from_host(rng.uniform(-1e-3, 1e-3, dhv.ndofs))
    res = dev.zeros(dhv.ndofs)
    P["calcium"]["field"] = rng.uniform(0.0, 0.6, (g.n_cells, 8))
    tb.update_linearization(op, ud, 0.0, residual=res)
    dev.synchronize()
    t_asm = t_rx = t_heat = t_lin = 0.0
    its = 0
    b = dev.zeros(npts)
    x = dev.zeros(npts)
    un_host = cache.un.to_host().reshape(model.nstates, npts)
    b.copy_from_host(un_host[model.phi_index])
    for k in range(args.reps):
        ev[0].record(); tb.update_operators(M, K, k * dt)
        ev[1].record(); tb.perform_step(f, cache, k * dt, dt)
        ev[2].record()
        tb.heat_system_matrix(dev, M, K, dt, A)
        rhs = dev.zeros(npts)
        M.mul(rhs, b)
        x.copy_from_host(un_host[model.phi_index])
        it, _ = tb.cg_solve(K.pattern, A, rhs, x, rtol=1e-6, atol=1e-8, maxiter=500)
        its += it
        ev[3].record(); tb.update_linearization(op, ud, k * dt, residual=res)
        ev[4].record()
        dev.synchronize()
        t_asm += ev[0].elapsed_ms(ev[1]); t_rx += ev[1].elapsed_ms(ev[2]); t_heat += ev[2].elapsed_ms(ev[3]); t_lin += ev[3].elapsed_ms(ev[4])
    R = args.reps
    nq = 8 if args.order == 1 else 27
    out = {"workload": "coupled electromechanics step on the ideal LV: %d hexahedra (nc %d, nr %d, nl %d), %d EP dofs x %d TT06 states, %d displacement dofs (order %d)"
                       % (g.n_cells, args.nc, args.nr, args.nl, npts, model.nstates, dhv.ndofs, args.order),
           "setup_s": setup_s,
           "ep_assembly_ms": t_asm / R, "ep_element_integrations_per_s": 2 * g.n_cells / (t_asm / R * 1e-3),
           "reaction_ms": t_rx / R, "reaction_dof_updates_per_s": model.nstates * npts / (t_rx / R * 1e-3),
           "heat_solve_ms": t_heat / R, "heat_cg_iterations": its / R,
           "mechanics_linearize_ms": t_lin / R, "mechanics_element_integrations_per_s": g.n_cells / (t_lin / R * 1e-3),
           "mechanics_quadrature_points_per_s": nq * g.n_cells / (t_lin / R * 1e-3),
           "patch_stats": K.pattern.patch_stats(),
           "note": "two kernel rates, no reference counterpart (SURVEY F6): EP = M + K (nodal fibre field, PATCH strategy on the unstructured LV mesh) + TT06 Rush-Larsen step + "
                   "BE heat solve (Jacobi-CG, rtol 1e-6); mechanics = active-stress Holzapfel-Ogden residual + tangent (element strategy)"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
