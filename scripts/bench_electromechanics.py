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


def build(tb, dev, nc, nr, nl, order=1, strategy="patch", part=None):
    """part = (world, rank): this rank's share of the ventricle under the recursive-bisection partition (tb.distributed.GeneralPartition) — the cells of
    the part as a local grid, their fibre fields, and for every neighbouring part the shared nodes (BASELINE configs[4] is an 8-GPU run)."""
    g_full = tb.generate_ideal_lv_mesh_hex(nc, nr, nl)
    f, s, n = tb.ideal_lv_microstructure(g_full, np.deg2rad(60.0), np.deg2rad(-60.0))
    apex = g_full.xyz[g_full.getnodeset("Apex")[0]].copy()
    g, gp = g_full, None
    if part is not None and part[0] > 1:
        cen = g_full.xyz[g_full.conn].mean(axis=1)
        gp = tb.distributed.GeneralPartition(g_full.conn, tb.distributed.partition_cells_rcb(cen, part[0]), part[1])
        g = tb.Grid(g_full.cell_kind, g_full.xyz[gp.global_nodes], gp.conn)
        f, s, n = f[gp.cells], s[gp.cells], n[gp.cells]
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
    return dict(g=g, g_full=g_full, gp=gp, apex=apex, fsn=(f, s, n), dhs=dhs, sps=sps, M=M, K=K, D=D, lam=lam, dhv=dhv, spv=spv, op=op, calcium=calcium)


def main_ranks(args):
    """--gpus N (N ≥ 1 under a launcher): one process per GPU, the ventricle bisected into N parts.  Every rank assembles the EP operators and the mechanics
    residual / tangent of its part with no data-path collective; the exchanges are the neighbour sums of the vectors (heat right-hand side, CG products,
    mechanics residual) and the CG's scalar all-reduces.  Timing: barrier-bracketed loop, max over ranks.  `checksums`: partition-independent sums (position-
    defined inputs, shared dofs weighted 1 / multiplicity) — an N-rank run prints the numbers of the one-rank run."""
    import torch
    import torch.distributed as dist
    import thunderbolt_jl_amd as tb
    rank, local_rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write("bench_electromechanics.py: WORLD_SIZE=%d but --gpus %d\n" % (world, args.gpus))
        return 2
    share = bool(os.environ.get("TB_BENCH_SHARE_DEVICE")) and world > 1      # tests on a one-GPU box: every rank on cuda:0 over gloo
    di = 0 if share else local_rank
    torch.cuda.set_device(di)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1 or "RANK" in os.environ:
        dist.init_process_group("gloo" if share else "nccl", **({} if share else {"device_id": torch.device("cuda", local_rank)}))
    gd = dist if dist.is_initialized() else None
    dev = tb.MI355XDevice(di)
    torch.cuda.set_stream(torch.cuda.Stream())
    dev.set_stream(torch.cuda.current_stream().cuda_stream)
    t0 = time.perf_counter()
    P = build(tb, dev, args.nc, args.nr, args.nl, args.order, part=(world, rank))
    g, gp, dhs, dhv, M, K, op = P["g"], P["gp"], P["dhs"], P["dhv"], P["M"], P["K"], P["op"]
    model = tb.TT06()
    npts, ndv = dhs.ndofs, dhv.ndofs
    n2d = tb.distributed.node_to_dof(dhs)
    nd0 = np.empty(g.n_nodes, dtype=np.int64)
    if args.order != 1:
        raise SystemExit("bench_electromechanics.py --gpus: first-order displacement (the shared-node lists are those of the grid nodes)")
    nd0[g.conn.ravel()] = dhv.cell_dofs[:, 0::3].ravel()
    nbr_s = [] if gp is None else [(q, torch.from_numpy(n2d[idx]).cuda()) for q, idx in gp.neighbours]
    nbr_v = [] if gp is None else [(q, torch.from_numpy((nd0[idx][:, None] + np.arange(3)).ravel()).cuda()) for q, idx in gp.neighbours]
    X = np.empty((npts, 3)); X[n2d] = g.xyz
    Xv = np.empty((ndv, 3))
    for c in range(3):
        Xv[nd0 + c] = g.xyz
    u0 = np.tile(model.default_initial_state(), (npts, 1)).T.copy()
    u0[model.phi_index, np.linalg.norm(X - P["apex"], axis=1) < 0.45] = 20.0
    f = tb.PointwiseODEFunction(npts, model)
    ut = torch.from_numpy(np.ascontiguousarray(u0).ravel()).cuda()
    cache = tb.setup_solver_cache(f, tb.RushLarsenCellSolver(dev), u=ut, keep_du=False)
    dt = 0.05
    wrap = lambda t: tb.DeviceVector.wrap(dev, t)                                                      # noqa: E731
    ud = torch.from_numpy(1e-3 * np.sin(3.0 * Xv[:, 0] + 2.0 * Xv[:, 1] + Xv[:, 2] + np.arange(ndv) % 3)).cuda()   # consistent on shared nodes: position and component only
    res = torch.zeros(ndv, dtype=torch.float64, device="cuda")
    P["calcium"]["field"] = 0.3 * (1.0 + np.sin(g.xyz[g.conn] @ np.array([1.0, 2.0, 3.0])))
    halo_s = tb.distributed.HaloExchange(nbr_s, gd, ut[:npts], dev)
    halo_v = tb.distributed.HaloExchange(nbr_v, gd, res, dev)
    b = torch.from_numpy(u0[model.phi_index].copy()).cuda()
    rhs, x = torch.zeros_like(b), torch.zeros_like(b)
    # warm-up: plans, tables, the heat matrix and the CG object of this partition
    tb.update_operators(M, K, 0.0)
    tb.perform_step(f, cache, 0.0, dt)
    A = tb.heat_system_matrix(dev, M, K, dt)
    diag = torch.empty(npts, dtype=torch.float64, device="cuda")
    tb._lib.check(tb.lib().tb_extract_diagonal(K.pattern.h, A.ptr, diag.data_ptr()))
    cg = tb.distributed.DistributedCG(None, diag, None, None, rank, world, gd, neighbours=nbr_s, device=dev, operator=(K.pattern, A))
    tb.update_linearization(op, wrap(ud), 0.0, residual=wrap(res))
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t0

    def step(k):
        tb.update_operators(M, K, k * dt)
        tb.perform_step(f, cache, k * dt, dt)
        tb.heat_system_matrix(dev, M, K, dt, A)
        M.mul(wrap(rhs), wrap(b))
        halo_s.exchange_sum(rhs)
        x.copy_(b)
        _, it, _ = cg.solve(rhs, x, rtol=1e-6, atol=1e-8, maxiter=500)
        tb.update_linearization(op, wrap(ud), k * dt, residual=wrap(res))
        halo_v.exchange_sum(res)
        return it

    step(0)
    torch.cuda.synchronize()
    if gd is not None:
        gd.barrier()
    t1 = time.perf_counter()
    its = 0
    for k in range(args.reps):
        its += step(k)
    torch.cuda.synchronize()
    if gd is not None:
        gd.barrier()
    el = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device="cpu" if share else "cuda")
    if gd is not None:
        gd.all_reduce(el, op=gd.ReduceOp.MAX)
    ms = float(el.item()) / args.reps * 1e3
    # checksums
    ws, wv = torch.ones(npts, dtype=torch.float64, device="cuda"), torch.ones(ndv, dtype=torch.float64, device="cuda")
    for _, idx in nbr_s:
        ws[idx] += 1.0
    for _, idx in nbr_v:
        wv[idx] += 1.0
    ws, wv = 1.0 / ws, 1.0 / wv
    psi = torch.from_numpy(np.cos(2.0 * X[:, 0] - X[:, 1]) + 0.5 * np.sin(4.0 * X[:, 2]) + 1.5).cuda()
    psv = torch.from_numpy(np.cos(2.0 * Xv[:, 0] - Xv[:, 1]) + 0.5 * np.sin(4.0 * Xv[:, 2]) + 1.5).cuda()
    y, yv = torch.empty_like(psi), torch.empty_like(psv)
    sums = []
    for o_ in (M, K):
        o_.mul(wrap(y), wrap(psi))
        sums.append((psi * y).sum())
    tb._lib.check(tb.lib().tb_spmv_csr(op.pattern.h, op.J.ptr, psv.data_ptr(), 1.0, 0.0, yv.data_ptr()))
    sums += [(psv * yv).sum(), (wv * res * res).sum(), (wv * res * psv).sum()]
    U = ut.view(model.nstates, npts)
    sums += [(ws * U * U).sum(), (ws * rhs * psi).sum()]
    t_ = torch.stack(sums)
    if world > 1:
        tb.distributed.all_reduce_sum(t_, gd)
    if rank == 0:
        names = ("psi_M_psi", "psi_K_psi", "psi_J_psi", "r_r", "r_psi", "u_u", "heat_rhs_psi")
        out = {"workload": "coupled electromechanics step on the ideal LV in %d part(s) (recursive bisection): %d hexahedra (nc %d, nr %d, nl %d), %d TT06 states per EP dof, first-order displacement"
                           % (world, P["g_full"].n_cells, args.nc, args.nr, args.nl, model.nstates),
               "n_gpus": world, "backend": None if gd is None else ("gloo (shared device, test)" if share else "nccl (RCCL)"),
               "cells_total": P["g_full"].n_cells, "cells_per_gpu": g.n_cells, "neighbour_parts": len(nbr_s), "shared_nodes": int(sum(len(i_) for _, i_ in nbr_s)),
               "setup_s": setup_s, "ms_per_step": ms, "heat_cg_iterations": its / args.reps,
               "element_integrations_per_s": 3 * P["g_full"].n_cells / (ms * 1e-3),
               "checksums": {k_: float(v_) for k_, v_ in zip(names, t_.cpu().numpy())},
               "note": "step = EP M + K (nodal fibre field, PATCH) + TT06 Rush-Larsen step + heat matrix + distributed Jacobi-CG heat solve (rtol 1e-6) + mechanics residual and tangent "
                       "(element strategy) + neighbour sums of the heat right-hand side and of the residual; 3 element-integrations per cell and step (M, K, mechanics)"}
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nc", type=int, default=128); ap.add_argument("--nr", type=int, default=8); ap.add_argument("--nl", type=int, default=100)
    ap.add_argument("--order", type=int, default=1, help="displacement order (1 or 2)")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--gpus", type=int, default=0, help="N ≥ 1: the partitioned form (one process per GPU, recursive bisection of the ventricle; under a launcher or spawned here)")
    args = ap.parse_args()
    if args.gpus >= 1:
        if "RANK" not in os.environ and args.gpus > 1:          # spawn the ranks (this process has not touched the GPU)
            import socket
            import subprocess
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                port = so.getsockname()[1]
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
                   "--master-port", str(port), "--", os.path.abspath(__file__)] + sys.argv[1:]
            return subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")))
        return main_ranks(args)
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
    sys.exit(main() or 0)
