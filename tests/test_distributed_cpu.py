"""N>1 path on CPU: world_size-2 `gloo` run of the slab partition + neighbour halo sum that bench.py uses over
RCCL.  The reference has no distributed code (README.md:7), so correctness is defined as
"P-rank result == 1-rank result" (SURVEY §8e).  Local assembly is done by the CPU oracle here (no GPU);
the partition / interface bookkeeping / exchange under test are the shipped ones."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nel, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import thunderbolt_jl_amd as tb
    from oracle import oracle as o
    D = tb.distributed
    left, right = (0.0, 0.0, 0.0), (1.0, 1.0, 2.0)
    part = D.SlabPartition(nel, left, right, world, rank)
    g = tb.generate_mesh(tb.Hexahedron, part.local_nel(), part.left, part.right)
    dh = tb.DofHandler(g)
    sp = tb.allocate_matrix(dh)
    om = o.Mesh(o.HEX8, 2, g.xyz, g.conn, dh.cell_dofs)
    n2d = D.node_to_dof(dh)
    lo, up = part.interface_nodes()
    lo_idx = None if lo is None else torch.from_numpy(n2d[lo])
    up_idx = None if up is None else torch.from_numpy(n2d[up])
    # source vector: local assembly, then halo sum
    b = torch.from_numpy(o.assemble_source(om, o.SRC_COS_EXP, t=0.1))
    # the persistent-buffer exchange (host tensors: torch index_select / index_add_ on buffers allocated once) equals the plain one bit for bit
    halo = D.HaloExchange(D.slab_neighbours(lo_idx, up_idx, rank, world), dist, b)
    b_packed = halo.exchange_sum(b.clone())
    D.halo_sum(b, lo_idx, up_idx, rank, world, dist)
    assert torch.equal(b, b_packed) and halo.nbytes == 8 * sum(i.numel() for i in (lo_idx, up_idx) if i is not None)
    # operator action y = K x with sub-domain (unassembled-at-the-interface) matrices: local SpMV + halo sum
    kap = np.diag([4.5e-5, 2.0e-5, 2.0e-5])
    Kp = o.assemble_matrix(om, 1, o.Coef(o.COEF_CONST_TENSOR, kap.ravel()), sp.rowptr, sp.colidx)
    xg = lambda X: np.sin(3 * X[:, 0]) + X[:, 1] ** 2 - 0.5 * X[:, 2]  # noqa: E731  (a global nodal field)
    xnode = xg(g.xyz)
    xl = np.empty(dh.ndofs)
    xl[n2d] = xnode
    y = torch.from_numpy(o.spmv_csr(sp.rowptr, sp.colidx, Kp, xl))
    D.halo_sum(y, lo_idx, up_idx, rank, world, dist)
    # report by global node id
    nx, ny, _ = nel
    plane = (nx + 1) * (ny + 1)
    gnode = np.arange(g.n_nodes) + part.z0 * plane
    q.put((rank, gnode, b.numpy()[n2d], y.numpy()[n2d]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nel", [(3, 2, 4), (4, 4, 5)])
def test_two_rank_halo_sum_equals_single_domain(nel):
    from oracle import oracle as o
    import thunderbolt_jl_amd as tb
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nel, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-domain reference
    g = tb.generate_mesh(tb.Hexahedron, nel, (0.0, 0.0, 0.0), (1.0, 1.0, 2.0))
    dh = tb.DofHandler(g)
    sp = tb.allocate_matrix(dh)
    om = o.Mesh(o.HEX8, 2, g.xyz, g.conn, dh.cell_dofs)
    n2d = tb.distributed.node_to_dof(dh)
    bref = o.assemble_source(om, o.SRC_COS_EXP, t=0.1)[n2d]
    K = o.assemble_matrix(om, 1, o.Coef(o.COEF_CONST_TENSOR, np.diag([4.5e-5, 2.0e-5, 2.0e-5]).ravel()), sp.rowptr, sp.colidx)
    x = np.empty(dh.ndofs)
    x[n2d] = np.sin(3 * g.xyz[:, 0]) + g.xyz[:, 1] ** 2 - 0.5 * g.xyz[:, 2]
    yref = o.spmv_csr(sp.rowptr, sp.colidx, K, x)[n2d]
    seen = np.zeros(g.n_nodes, dtype=int)
    for rank, gnode, b, y in res:
        # local slab coordinates must coincide with the global lattice
        np.testing.assert_allclose(b, bref[gnode], rtol=1e-12, atol=1e-18)
        np.testing.assert_allclose(y, yref[gnode], rtol=1e-10, atol=1e-16)
        seen[gnode] += 1
    assert seen.min() == 1 and seen.max() == 2  # interface plane is held by both ranks
    plane = (nel[0] + 1) * (nel[1] + 1)
    assert (seen == 2).sum() == plane


def _cg_worker(rank, world, port, nel, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import thunderbolt_jl_amd as tb
    from oracle import oracle as o
    D = tb.distributed
    part = D.SlabPartition(nel, (0.0, 0.0, 0.0), (1.0, 1.0, 2.0), world, rank)
    g = tb.generate_mesh(tb.Hexahedron, part.local_nel(), part.left, part.right)
    dh = tb.DofHandler(g)
    sp = tb.allocate_matrix(dh)
    om = o.Mesh(o.HEX8, 2, g.xyz, g.conn, dh.cell_dofs)
    n2d = D.node_to_dof(dh)
    lo, up = part.interface_nodes()
    lo_idx = None if lo is None else torch.from_numpy(n2d[lo])
    up_idx = None if up is None else torch.from_numpy(n2d[up])
    kap = np.diag([4.5e-2, 2.0e-2, 2.0e-2])
    Mp = o.assemble_matrix(om, 0, o.Coef(o.COEF_CONST_SCALAR, [1.0]), sp.rowptr, sp.colidx)
    Kp = o.assemble_matrix(om, 1, o.Coef(o.COEF_CONST_TENSOR, kap.ravel()), sp.rowptr, sp.colidx)
    Ap = o.heat_matrix(Mp, Kp, 0.5)
    diag = np.array([Ap[sp.rowptr[r] + np.searchsorted(sp.colidx[sp.rowptr[r]:sp.rowptr[r + 1]], r)] for r in range(dh.ndofs)])
    spmv = lambda x: torch.from_numpy(o.spmv_csr(sp.rowptr, sp.colidx, Ap, x.numpy()))  # noqa: E731
    cg = D.DistributedCG(spmv, torch.from_numpy(diag), lo_idx, up_idx, rank, world, dist)
    u0 = np.empty(dh.ndofs)
    u0[n2d] = np.cos(2 * g.xyz[:, 0]) * (1 + g.xyz[:, 2])            # consistent initial state (global nodal field)
    b = torch.from_numpy(o.spmv_csr(sp.rowptr, sp.colidx, Mp, u0))   # b = M uₙ₋₁ (local) …
    D.halo_sum(b, lo_idx, up_idx, rank, world, dist)                 # … assembled over the interface
    x, its, rn = cg.solve(b, torch.from_numpy(u0.copy()), rtol=1e-12, atol=1e-14)
    plane = (nel[0] + 1) * (nel[1] + 1)
    q.put((rank, np.arange(g.n_nodes) + part.z0 * plane, x.numpy()[n2d], its))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_heat_solve_equals_single_domain():
    import scipy.sparse as sps
    import scipy.sparse.linalg as spla
    from oracle import oracle as o
    import thunderbolt_jl_amd as tb
    nel, world = (4, 3, 6), 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cg_worker, args=(r, world, port, nel, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = tb.generate_mesh(tb.Hexahedron, nel, (0.0, 0.0, 0.0), (1.0, 1.0, 2.0))
    dh = tb.DofHandler(g)
    sp = tb.allocate_matrix(dh)
    om = o.Mesh(o.HEX8, 2, g.xyz, g.conn, dh.cell_dofs)
    n2d = tb.distributed.node_to_dof(dh)
    n = dh.ndofs
    M = o.assemble_matrix(om, 0, o.Coef(o.COEF_CONST_SCALAR, [1.0]), sp.rowptr, sp.colidx)
    K = o.assemble_matrix(om, 1, o.Coef(o.COEF_CONST_TENSOR, np.diag([4.5e-2, 2.0e-2, 2.0e-2]).ravel()), sp.rowptr, sp.colidx)
    csr = lambda nz: sps.csr_matrix((nz, sp.colidx, sp.rowptr), shape=(n, n))  # noqa: E731
    u0 = np.empty(n)
    u0[n2d] = np.cos(2 * g.xyz[:, 0]) * (1 + g.xyz[:, 2])
    ref = spla.spsolve(csr(o.heat_matrix(M, K, 0.5)).tocsc(), csr(M) @ u0)[n2d]
    for rank, gnode, x, its in res:
        assert 0 < its < 200
        np.testing.assert_allclose(x, ref[gnode], rtol=1e-9, atol=1e-12)


def _unstructured_problem(tb):
    """A perturbed box presented as an unstructured mesh: cells shuffled, nodes renumbered at random."""
    rng = np.random.default_rng(11)
    g0 = tb.generate_mesh(tb.Hexahedron, (5, 4, 6), (0.0, 0.0, 0.0), (1.0, 0.8, 1.4), perturb=0.2)
    perm = rng.permutation(g0.n_nodes)
    xyz = np.empty_like(g0.xyz)
    xyz[perm] = g0.xyz
    conn = perm[g0.conn][rng.permutation(g0.n_cells)].astype(np.int32)
    return xyz, conn


def _general_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import thunderbolt_jl_amd as tb
    from oracle import oracle as o
    D = tb.distributed
    xyz, conn = _unstructured_problem(tb)
    part = D.partition_cells_rcb(xyz[conn].mean(axis=1), world)
    gp = D.GeneralPartition(conn, part, rank)
    lx = xyz[gp.global_nodes]
    cd, nd = o.close_dofs(o.HEX8, 1, gp.conn, len(lx))
    rp, ci = o.build_pattern(cd, nd)
    om = o.Mesh(o.HEX8, 2, lx, gp.conn, cd)
    n2d = np.full(len(lx), -1, dtype=np.int64)
    n2d[gp.conn.ravel()] = cd.ravel()
    nb = [(peer, torch.from_numpy(n2d[idx])) for peer, idx in gp.neighbours]
    b = torch.from_numpy(o.assemble_source(om, o.SRC_COS_EXP, t=0.1))
    D.exchange_sum(b, nb, dist)
    kap = np.diag([4.5e-2, 2.0e-2, 2.0e-2])
    Mp = o.assemble_matrix(om, 0, o.Coef(o.COEF_CONST_SCALAR, [1.0]), rp, ci)
    Kp = o.assemble_matrix(om, 1, o.Coef(o.COEF_CONST_TENSOR, kap.ravel()), rp, ci)
    Ap = o.heat_matrix(Mp, Kp, 0.5)
    diag = np.array([Ap[rp[r] + np.searchsorted(ci[rp[r]:rp[r + 1]], r)] for r in range(nd)])
    cg = D.DistributedCG(lambda x: torch.from_numpy(o.spmv_csr(rp, ci, Ap, x.numpy())), torch.from_numpy(diag), None, None, rank, world, dist,
                         neighbours=nb)
    u0 = np.empty(nd)
    u0[n2d] = np.cos(2 * lx[:, 0]) * (1 + lx[:, 2])
    rhs = torch.from_numpy(o.spmv_csr(rp, ci, Mp, u0))
    D.exchange_sum(rhs, nb, dist)
    x, its, rn = cg.solve(rhs, torch.from_numpy(u0.copy()), rtol=1e-12, atol=1e-14)
    q.put((rank, gp.global_nodes, b.numpy()[n2d], x.numpy()[n2d], len(gp.cells), gp.multiplicity()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_general_partition_equals_single_domain(world):
    """Unstructured presentation, recursive-coordinate-bisection parts (3 parts: some nodes are held by three ranks):
    assembled source vector and the backward-Euler heat solve equal the single-domain result."""
    import scipy.sparse as sps
    import scipy.sparse.linalg as spla
    from oracle import oracle as o
    import thunderbolt_jl_amd as tb
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_general_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    xyz, conn = _unstructured_problem(tb)
    cd, nd = o.close_dofs(o.HEX8, 1, conn, len(xyz))
    rp, ci = o.build_pattern(cd, nd)
    om = o.Mesh(o.HEX8, 2, xyz, conn, cd)
    n2d = np.full(len(xyz), -1, dtype=np.int64)
    n2d[conn.ravel()] = cd.ravel()
    bref = o.assemble_source(om, o.SRC_COS_EXP, t=0.1)[n2d]
    M = o.assemble_matrix(om, 0, o.Coef(o.COEF_CONST_SCALAR, [1.0]), rp, ci)
    K = o.assemble_matrix(om, 1, o.Coef(o.COEF_CONST_TENSOR, np.diag([4.5e-2, 2.0e-2, 2.0e-2]).ravel()), rp, ci)
    A = sps.csr_matrix((o.heat_matrix(M, K, 0.5), ci, rp), shape=(nd, nd))
    u0 = np.empty(nd)
    u0[n2d] = np.cos(2 * xyz[:, 0]) * (1 + xyz[:, 2])
    xref = spla.spsolve(A.tocsc(), sps.csr_matrix((M, ci, rp), shape=(nd, nd)) @ u0)[n2d]
    held = np.zeros(len(xyz), dtype=int)
    ncells = 0
    for rank, gnodes, b, x, nc, mult in res:
        np.testing.assert_allclose(b, bref[gnodes], rtol=1e-12, atol=1e-18)
        np.testing.assert_allclose(x, xref[gnodes], rtol=1e-9, atol=1e-12)
        held[gnodes] += 1
        ncells += nc
    assert ncells == len(conn) and held.min() == 1
    for rank, gnodes, b, x, nc, mult in res:
        np.testing.assert_array_equal(mult, held[gnodes])          # each rank knows how many ranks hold each of its nodes
    if world == 3:
        assert held.max() == 3
    sizes = sorted(r[4] for r in res)
    assert sizes[-1] - sizes[0] <= 1                               # balanced parts


def _mechanics_field(x):
    """a smooth displacement as a function of position, so that every rank evaluates the same field on its own nodes"""
    return 1e-2 * np.stack([np.sin(2 * x[:, 1]) * x[:, 2], np.cos(x[:, 0]) * x[:, 1], x[:, 0] * x[:, 1] - 0.5 * x[:, 2]], axis=1)


def _mechanics_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import thunderbolt_jl_amd as tb
    from oracle import oracle as o
    D = tb.distributed
    xyz, conn = _unstructured_problem(tb)
    part = D.partition_cells_rcb(xyz[conn].mean(axis=1), world)
    gp = D.GeneralPartition(conn, part, rank)
    lx = xyz[gp.global_nodes]
    cd, nd = o.close_dofs(o.HEX8, 3, gp.conn, len(lx))           # three displacement dofs per node, node-major
    rp, ci = o.build_pattern(cd, nd)
    om = o.Mesh(o.HEX8, 2, lx, gp.conn, cd)
    n2d0 = np.full(len(lx), -1, dtype=np.int64)
    n2d0[gp.conn.ravel()] = cd[:, 0::3].ravel()
    u = np.empty(nd)
    uf = _mechanics_field(lx)
    for c in range(3):
        u[n2d0 + c] = uf[:, c]
    fsn = np.eye(3)
    K, r = o.assemble_hyperelastic(om, u, rp, ci, fsn=fsn)
    # the interface exchange of a vector field: the three dofs of every shared node, in the shared order of the nodes
    nb = [(peer, torch.from_numpy((n2d0[idx][:, None] + np.arange(3)).ravel())) for peer, idx in gp.neighbours]
    rt = torch.from_numpy(r.copy())
    D.exchange_sum(rt, nb, dist)
    # K·v with a global v: local product, then the same exchange (sub-domain matrices are never summed across ranks)
    v = np.empty(nd)
    vf = np.stack([np.cos(3 * lx[:, 0]), lx[:, 1] ** 2, np.sin(lx[:, 2] + lx[:, 0])], axis=1)
    for c in range(3):
        v[n2d0 + c] = vf[:, c]
    Kv = torch.from_numpy(o.spmv_csr(rp, ci, K, v))
    D.exchange_sum(Kv, nb, dist)
    q.put((rank, gp.global_nodes, np.stack([rt.numpy()[n2d0 + c] for c in range(3)], axis=1), np.stack([Kv.numpy()[n2d0 + c] for c in range(3)], axis=1)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_distributed_mechanics_residual_and_tangent_action(world):
    """SURVEY §8e for the mechanics path: cells partitioned by recursive bisection, every rank assembles the hyperelastic residual and
    tangent of its own cells only, one interface sum over the three dofs of every shared node — the residual and the action K·v equal
    the single-domain ones to 1e-12 (only the summation order on interface dofs differs)."""
    from oracle import oracle as o
    import thunderbolt_jl_amd as tb
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mechanics_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    xyz, conn = _unstructured_problem(tb)
    cd, nd = o.close_dofs(o.HEX8, 3, conn, len(xyz))
    rp, ci = o.build_pattern(cd, nd)
    om = o.Mesh(o.HEX8, 2, xyz, conn, cd)
    n2d0 = np.full(len(xyz), -1, dtype=np.int64)
    n2d0[conn.ravel()] = cd[:, 0::3].ravel()
    u, v = np.empty(nd), np.empty(nd)
    uf = _mechanics_field(xyz)
    vf = np.stack([np.cos(3 * xyz[:, 0]), xyz[:, 1] ** 2, np.sin(xyz[:, 2] + xyz[:, 0])], axis=1)
    for c in range(3):
        u[n2d0 + c] = uf[:, c]; v[n2d0 + c] = vf[:, c]
    K, r = o.assemble_hyperelastic(om, u, rp, ci, fsn=np.eye(3))
    Kv = o.spmv_csr(rp, ci, K, v)
    rref = np.stack([r[n2d0 + c] for c in range(3)], axis=1)
    Kvref = np.stack([Kv[n2d0 + c] for c in range(3)], axis=1)
    for rank, gnodes, rr, kk in res:
        np.testing.assert_allclose(rr, rref[gnodes], rtol=1e-11, atol=1e-12 * np.abs(rref).max())
        np.testing.assert_allclose(kk, Kvref[gnodes], rtol=1e-11, atol=1e-12 * np.abs(Kvref).max())


# ----------------------------------------------------------------------------------------------- BASELINE config 5 under a partition
def _lv_problem(tb):
    """idealised left ventricle (all-hexahedral, O-grid apex), rule-based fibres, a smooth calcium-like activation field of position"""
    g = tb.generate_ideal_lv_mesh_hex(16, 2, 8)
    f, s, n = tb.ideal_lv_microstructure(g, np.deg2rad(60.0), np.deg2rad(-60.0))
    return g, np.stack([f, s, n], axis=2)                      # fsn[cell][basis][f|s|n][3]


def _lv_fields(x):
    phi = -80.0 + 100.0 * np.exp(-2.0 * ((x[:, 0] - 0.2) ** 2 + x[:, 1] ** 2 + (x[:, 2] + 0.8) ** 2))
    act = 0.5 + 0.4 * np.sin(2.0 * x[:, 0]) * np.cos(1.5 * x[:, 2])
    disp = 5e-3 * np.stack([np.sin(2 * x[:, 1]) * x[:, 2], np.cos(x[:, 0]) * x[:, 1], x[:, 0] * x[:, 1] - 0.5 * x[:, 2]], axis=1)
    return phi, act, disp


def _lv_local(o, xyz, conn, fsn, lam, tmax):
    """the two halves of a coupled step on one (sub-)mesh with the oracle: fibre-aligned K·φ and M·φ, a TT06 Rush–Larsen step of the points, the
    active-stress residual and tangent action; returns per-NODE arrays"""
    phi, act, disp = _lv_fields(xyz)
    cd, nd = o.close_dofs(o.HEX8, 1, conn, len(xyz))
    rp, ci = o.build_pattern(cd, nd)
    om = o.Mesh(o.HEX8, 2, xyz, conn, cd)
    n2d = np.full(len(xyz), -1, dtype=np.int64)
    n2d[conn.ravel()] = cd.ravel()
    K = o.assemble_matrix(om, 1, o.Coef(o.COEF_SPECTRAL_FIELD, lam, field=np.ascontiguousarray(fsn), Cm=1.0, chi=1.0, wrap=True), rp, ci)
    M = o.assemble_matrix(om, 0, o.Coef(o.COEF_CONST_SCALAR, [1.0]), rp, ci)
    x = np.empty(nd); x[n2d] = phi
    Kx, Mx = o.spmv_csr(rp, ci, K, x), o.spmv_csr(rp, ci, M, x)
    # reaction: the points of this (sub-)mesh, SoA
    p = o.cell_default_params(o.CELL_TT06)
    u = np.tile(o.cell_default_state(o.CELL_TT06, p), (nd, 1)).T.copy()
    u[0] = x
    u = np.ascontiguousarray(u).ravel()
    o.reaction_step_rl(o.CELL_TT06, p, u, nd, o.LAYOUT_SOA, t=0.0, dt=0.02)
    un = u.reshape(-1, nd)[:, n2d]                                     # states per node
    # mechanics: three dofs per node, activation per cell and node
    cdv, ndv = o.close_dofs(o.HEX8, 3, conn, len(xyz))
    rpv, civ = o.build_pattern(cdv, ndv)
    omv = o.Mesh(o.HEX8, 2, xyz, conn, cdv)
    n2d0 = np.full(len(xyz), -1, dtype=np.int64)
    n2d0[conn.ravel()] = cdv[:, 0::3].ravel()
    uv, vv = np.empty(ndv), np.empty(ndv)
    vf = np.stack([np.cos(3 * xyz[:, 0]), xyz[:, 1] ** 2, np.sin(xyz[:, 2] + xyz[:, 0])], axis=1)
    for c in range(3):
        uv[n2d0 + c] = disp[:, c]; vv[n2d0 + c] = vf[:, c]
    o.set_microstructure_field(np.ascontiguousarray(fsn))
    o.set_active_tension(tmax, np.ascontiguousarray(act[conn]))
    try:
        Kt, r = o.assemble_hyperelastic(omv, uv, rpv, civ)
    finally:
        o.set_active_tension(0.0); o.set_microstructure_field(None)
    Kv = o.spmv_csr(rpv, civ, Kt, vv)
    return dict(n2d=n2d, n2d0=n2d0, Kx=Kx, Mx=Mx, un=un, r=r, Kv=Kv)


def _lv_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import thunderbolt_jl_amd as tb
    from oracle import oracle as o
    D = tb.distributed
    g, fsn = _lv_problem(tb)
    part = D.partition_cells_rcb(g.xyz[g.conn].mean(axis=1), world)
    gp = D.GeneralPartition(g.conn, part, rank)
    lx = g.xyz[gp.global_nodes]
    L = _lv_local(o, lx, gp.conn, fsn[gp.cells], np.array([0.3, 0.12, 0.12]) * 1e-2, 20.0)
    nb = [(peer, torch.from_numpy(L["n2d"][idx])) for peer, idx in gp.neighbours]
    nb3 = [(peer, torch.from_numpy((L["n2d0"][idx][:, None] + np.arange(3)).ravel())) for peer, idx in gp.neighbours]
    Kx, Mx = torch.from_numpy(L["Kx"].copy()), torch.from_numpy(L["Mx"].copy())
    D.exchange_sum(Kx, nb, dist); D.exchange_sum(Mx, nb, dist)
    r, Kv = torch.from_numpy(L["r"].copy()), torch.from_numpy(L["Kv"].copy())
    D.exchange_sum(r, nb3, dist); D.exchange_sum(Kv, nb3, dist)
    n2d, n2d0 = L["n2d"], L["n2d0"]
    q.put((rank, gp.global_nodes, Kx.numpy()[n2d], Mx.numpy()[n2d], L["un"], np.stack([r.numpy()[n2d0 + c] for c in range(3)], axis=1),
           np.stack([Kv.numpy()[n2d0 + c] for c in range(3)], axis=1), len(gp.cells)))
    dist.barrier()
    dist.destroy_process_group()


def test_config5_lv_coupled_step_under_a_partition():
    """BASELINE config 5 (coupled electromechanics on the ventricle mesh, 8 GPUs) has no reference counterpart; its multi-GPU form is defined as
    "P-rank result == 1-rank result" (SURVEY §8e).  Two ranks, recursive-coordinate-bisection parts of the idealised LV: every rank assembles the
    fibre-aligned monodomain operators and the active-stress mechanics residual / tangent of its own cells only (the oracle stands in for the device
    kernels, as in the tests above), steps the TT06 states of its own points with no communication, and the interface sums (scalar dofs; the three
    displacement dofs of every shared node) make K·φ, M·φ, the residual and the tangent action equal to the single-domain ones."""
    from oracle import oracle as o
    import thunderbolt_jl_amd as tb
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_lv_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g, fsn = _lv_problem(tb)
    R = _lv_local(o, g.xyz, g.conn, fsn, np.array([0.3, 0.12, 0.12]) * 1e-2, 20.0)
    n2d, n2d0 = R["n2d"], R["n2d0"]
    Kx, Mx = R["Kx"][n2d], R["Mx"][n2d]
    rr = np.stack([R["r"][n2d0 + c] for c in range(3)], axis=1)
    kv = np.stack([R["Kv"][n2d0 + c] for c in range(3)], axis=1)
    held = np.zeros(g.n_nodes, dtype=int)
    ncells = 0
    for rank, gnodes, kx, mx, un, r3, kv3, nc in res:
        np.testing.assert_allclose(kx, Kx[gnodes], rtol=1e-11, atol=1e-12 * np.abs(Kx).max())
        np.testing.assert_allclose(mx, Mx[gnodes], rtol=1e-11, atol=1e-12 * np.abs(Mx).max())
        np.testing.assert_array_equal(un, R["un"][:, gnodes])                       # the reaction step is pointwise: identical bits on every rank that holds the point
        np.testing.assert_allclose(r3, rr[gnodes], rtol=1e-10, atol=1e-11 * np.abs(rr).max())
        np.testing.assert_allclose(kv3, kv[gnodes], rtol=1e-10, atol=1e-11 * np.abs(kv).max())
        held[gnodes] += 1
        ncells += nc
    assert ncells == g.n_cells and held.min() == 1 and held.max() == 2
