"""N>1 path with device-assembled operators: two ranks, both on cuda:0, `gloo` backend with halo buffers staged through the host (the box has
one GPU; RCCL needs one device per rank).  Every rank assembles b, M and K of its z-slab with the HIP kernels (the fused M + K pass and the
vector scatter bench.py runs), and the partition / interface bookkeeping / exchange / distributed Jacobi-CG under test are the shipped ones
(thunderbolt.jl_amd/distributed.py, device path: tb_spmv_csr / tb_spmv_csr_dot / tb_spmv_csr_rows + tb_cgd_* with device-resident scalars, halo pack /
unpack through tb_gather_indexed / tb_scatter_add_indexed).  The reference is shared-memory only
(README.md:7): correctness is "P-rank result == 1-rank result" (SURVEY §8e), here against the 1-rank HIP result."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NEL = (12, 10, 16)
LEFT, RIGHT = (0.0, 0.0, 0.0), (1.0, 1.0, 2.0)
KAPPA = np.diag([4.5e-2, 2.0e-2, 2.0e-2])
DT = 0.5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _warp(xyz):
    """smooth distortion of the global lattice as a function of position only, so slabs and the whole box see the same non-affine cells"""
    x, y, z = xyz[:, 0].copy(), xyz[:, 1].copy(), xyz[:, 2].copy()
    s = 0.012 * np.sin(2 * np.pi * x) * np.sin(2 * np.pi * y) * np.sin(np.pi * z)
    xyz[:, 0] += s
    xyz[:, 1] -= 0.7 * s
    xyz[:, 2] += 0.5 * s
    return xyz


def _field(X):
    return np.cos(2 * X[:, 0]) * (1 + X[:, 2]) + X[:, 1] ** 2


def _assemble(tb, dev, g):
    import torch
    dh = tb.DofHandler(g)
    sp = tb.allocate_matrix(dh)
    st = tb.PatchAssemblyStrategy(dev)
    M = tb.setup_operator(st, tb.BilinearMassIntegrator(tb.ConstantCoefficient(1.0)), dh, sp)
    K = tb.setup_operator(st, tb.BilinearDiffusionIntegrator(tb.ConstantCoefficient(KAPPA)), dh, sp)
    tb.update_operators(M, K, 0.0)
    src = tb.setup_operator(tb.AtomicAssemblyStrategy(dev), tb.LinearIntegrator(tb.AnalyticalCoefficient("cos_exp")), dh)
    b = torch.zeros(dh.ndofs, dtype=torch.float64, device="cuda")
    src.b = tb.DeviceVector.wrap(dev, b)
    tb.update_operator(src, 0.1)
    A = tb.heat_system_matrix(dev, M, K, DT)
    return dh, sp, M, K, A, b


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import thunderbolt_jl_amd as tb
    D = tb.distributed
    torch.cuda.set_device(0)
    dev = tb.MI355XDevice(0)
    dev.set_stream(torch.cuda.current_stream().cuda_stream)
    part = D.SlabPartition(NEL, LEFT, RIGHT, world, rank)
    g = tb.generate_mesh(tb.Hexahedron, part.local_nel(), part.left, part.right)
    _warp(g.xyz)
    dh, sp, M, K, A, b = _assemble(tb, dev, g)
    n2d = D.node_to_dof(dh)
    lo, up = part.interface_nodes()
    lo_idx = None if lo is None else torch.from_numpy(n2d[lo]).cuda()
    up_idx = None if up is None else torch.from_numpy(n2d[up]).cuda()
    torch.cuda.synchronize()
    # the packed exchange behind the C ABI (persistent buffers, tb_gather_indexed / tb_scatter_add_indexed) against plain torch indexing: bit for bit
    nb = D.slab_neighbours(lo_idx, up_idx, rank, world)
    halo = D.HaloExchange(nb, dist, b, dev)
    b_torch = D.halo_sum(b.clone(), lo_idx, up_idx, rank, world, dist)
    halo.exchange_sum(b)
    packed_equal = bool(torch.equal(b, b_torch))
    b_again = b_torch.clone()
    halo.exchange_sum(b_again)                                   # the buffers are reused: a second exchange on other data
    packed_equal = packed_equal and bool(torch.equal(b_again, D.halo_sum(b_torch.clone(), lo_idx, up_idx, rank, world, dist)))
    # operator action with the sub-domain matrix: device SpMV + exchange
    xl = np.empty(dh.ndofs)
    xl[n2d] = _field(g.xyz)
    x = torch.from_numpy(xl).cuda()
    y = torch.zeros_like(x)
    K.mul(tb.DeviceVector.wrap(dev, y), tb.DeviceVector.wrap(dev, x))
    D.halo_sum(y, lo_idx, up_idx, rank, world, dist)
    # backward-Euler heat step (M − Δt K) u = M u₀ by the distributed device CG
    Ah = A.to_host()
    diag = np.array([Ah[sp.rowptr[r] + np.searchsorted(sp.colidx[sp.rowptr[r]:sp.rowptr[r + 1]], r)] for r in range(dh.ndofs)])

    def spmv(v):
        out = torch.zeros_like(v)
        tb._lib.check(tb.lib().tb_spmv_csr(K.pattern.h, A.ptr, v.data_ptr(), 1.0, 0.0, out.data_ptr()))
        return out

    rhs = torch.zeros_like(x)
    M.mul(tb.DeviceVector.wrap(dev, rhs), tb.DeviceVector.wrap(dev, x))
    D.halo_sum(rhs, lo_idx, up_idx, rank, world, dist)
    cg = D.DistributedCG(spmv, torch.from_numpy(diag).cuda(), lo_idx, up_idx, rank, world, dist, device=dev, look=4)
    u, its, rn = cg.solve(rhs, x.clone(), rtol=1e-13, atol=1e-15, maxiter=400)
    # the same solve with the iteration ordered for overlap: interface rows of A·p packed first (tb_spmv_csr_rows), exchange posted, whole local
    # product + local pᵀA_p p behind it (tb_spmv_csr_dot), received partials added (tb_scatter_add_indexed)
    cg2 = D.DistributedCG(None, torch.from_numpy(diag).cuda(), lo_idx, up_idx, rank, world, dist, device=dev, look=4, operator=(K.pattern, A))
    u2, its2, rn2 = cg2.solve(rhs, x.clone(), rtol=1e-13, atol=1e-15, maxiter=400)
    # one product of each form on the same vector: the overlapped form packs rows computed by another kernel (summation order within a row
    # differs from the stream SpMV's), so agreement is to rounding, not bitwise
    S = torch.zeros(6, dtype=torch.float64, device="cuda")
    Ap1, Ap2 = torch.empty_like(x), torch.empty_like(x)
    cg.device_iteration(x, Ap1, S)
    pAp1 = float(S[1].item()); S.zero_()
    cg2.device_iteration(x, Ap2, S)
    pAp2 = float(S[1].item())
    prod_err = float((Ap1 - Ap2).abs().max() / Ap1.abs().max())
    # interface rows of the overlapped product: own rows + received rows, both from tb_spmv_csr_rows → the two sides of the plane hold the same bits
    ap_lo = None if lo_idx is None else Ap2[lo_idx].cpu().numpy()
    ap_up = None if up_idx is None else Ap2[up_idx].cpu().numpy()
    # an indefinite operator must be reported, not iterated to maxiter
    cgn = D.DistributedCG(lambda v: -spmv(v), torch.from_numpy(diag).cuda(), lo_idx, up_idx, rank, world, dist, device=dev, look=2)
    try:
        cgn.solve(rhs, x.clone(), rtol=1e-13, atol=0.0, maxiter=50)
        breakdown_reported = False
    except ArithmeticError:
        breakdown_reported = True
    torch.cuda.synchronize()
    plane = (NEL[0] + 1) * (NEL[1] + 1)
    gnode = np.arange(g.n_nodes) + part.z0 * plane
    q.put((rank, gnode, b.cpu().numpy()[n2d], y.cpu().numpy()[n2d], u.cpu().numpy()[n2d], its,
           {"packed_equal": packed_equal, "u2": u2.cpu().numpy()[n2d], "its2": its2, "prod_err": prod_err, "pAp": (pAp1, pAp2), "breakdown_reported": breakdown_reported,
            "ap_lo": ap_lo, "ap_up": ap_up}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])   # three ranks: the middle one has two neighbours — HaloExchange's merged gather / scatter-add, product rows packed into slices
def test_ranks_hip_assembly_halo_and_cg_equal_single_rank(tb, device, world):
    import torch
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # single-rank HIP reference on the whole box
    g = tb.generate_mesh(tb.Hexahedron, NEL, LEFT, RIGHT)
    _warp(g.xyz)
    dh, sp, M, K, A, b = _assemble(tb, device, g)
    n2d = tb.distributed.node_to_dof(dh)
    bref = b.cpu().numpy()[n2d]
    xl = np.empty(dh.ndofs)
    xl[n2d] = _field(g.xyz)
    x = device.to_device(xl)
    y = device.zeros(dh.ndofs)
    K.mul(y, x)
    yref = y.to_host()[n2d]
    rhs = device.zeros(dh.ndofs)
    M.mul(rhs, x)
    u = device.to_device(xl)
    its, _ = tb.cg_solve(K.pattern, A, rhs, u, rtol=1e-13, atol=1e-15, maxiter=400)
    uref = u.to_host()[n2d]
    seen = np.zeros(g.n_nodes, dtype=int)
    for rank, gnode, bb, yy, uu, it, extra in res:
        np.testing.assert_allclose(bb, bref[gnode], rtol=1e-12, atol=1e-12 * np.abs(bref).max())
        np.testing.assert_allclose(yy, yref[gnode], rtol=0, atol=1e-12 * np.abs(yref).max())
        np.testing.assert_allclose(uu, uref[gnode], rtol=0, atol=1e-9 * np.abs(uref).max())
        assert 0 < it <= 400
        assert extra["packed_equal"], "packed halo exchange (C ABI) differs from the torch-indexing exchange"
        np.testing.assert_allclose(extra["u2"], uref[gnode], rtol=0, atol=1e-9 * np.abs(uref).max())   # overlapped iteration, same solution
        assert 0 < extra["its2"] <= 400 and extra["prod_err"] < 1e-13
        np.testing.assert_allclose(extra["pAp"][0], extra["pAp"][1], rtol=1e-12)
        assert extra["breakdown_reported"]
        seen[gnode] += 1
    plane = (NEL[0] + 1) * (NEL[1] + 1)
    assert seen.min() == 1 and (seen == 2).sum() == plane * (world - 1)
    by_rank = {rank: extra for rank, _, _, _, _, _, extra in res}
    for r in range(world - 1):
        np.testing.assert_array_equal(by_rank[r]["ap_up"], by_rank[r + 1]["ap_lo"])   # bitwise consistent across every interface


def test_bench_refuses_wrong_job_size():
    """`python bench.py --gpus 2` on a one-GPU box must fail loudly instead of printing an n_gpus: 1 line."""
    import subprocess
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 GPUs")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "n_gpus" not in r.stdout


def test_bench_two_ranks_end_to_end_on_one_device():
    """bench.py's own N > 1 path — partition, halo sum of b between the ranks, barrier-bracketed timing, max over ranks, one JSON line from rank 0 —
    run as the driver runs it (`python bench.py --gpus 2` spawning its ranks), with both ranks on cuda:0 over gloo (TB_BENCH_SHARE_DEVICE, the
    test-only switch: the box has one GPU and RCCL wants one device per rank)."""
    import json
    import subprocess
    env = dict(os.environ, TB_BENCH_SHARE_DEVICE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--n", "24", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    assert "NaN" not in lines[0] and "Infinity" not in lines[0]      # strict JSON: every phase of an N-rank line is a number (--phase-events auto fills them)
    assert [ln for ln in r.stdout.splitlines() if ln.strip()][-1] == lines[0]   # and it is the last thing on stdout
    d = json.loads(lines[0])
    # default = strong scaling: the SAME 24³ mesh in two z-slabs of 12 layers (BASELINE's 1/2/4/8-GPU configuration)
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["cells_total"] == 24 ** 3 and d["config"]["cells_per_gpu"] == 24 * 24 * 12 and d["config"]["layers_per_gpu"] == 12
    assert d["config"]["backend"].startswith("gloo")
    assert d["phase_ms"]["halo"] > 0.0
    assert d["cg_iterations_per_s"] > 0 and d["distributed_cg"]["halo_bytes_per_rank"] == 25 * 25 * 8
    # value counts the mesh once: 3 element-integrations per cell of the 24³ mesh per step
    assert abs(d["value"] - 3 * 24 ** 3 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]


def test_bench_two_ranks_weak_mode_still_available():
    import json
    import subprocess
    env = dict(os.environ, TB_BENCH_SHARE_DEVICE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scaling", "weak", "--n", "16", "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
                        "--no-dist-cg"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["scaling"] == "weak" and d["config"]["cells_per_gpu"] == 16 ** 3 and d["config"]["cells_total"] == 2 * 16 ** 3


def _bench_line(args, env=None, timeout=900):
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=dict(os.environ, **(env or {})), capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("mesh_args", [["--mesh", "lv", "--lv-dims", "24,4,16"], ["--mesh", "shuffled", "--n", "16"], ["--mesh", "lv", "--lv-dims", "24,4,16", "--renumber", "grid"]])
def test_bench_unstructured_meshes_under_the_bisection_partition_match_one_rank(mesh_args):
    """BASELINE configs[4] is an 8-GPU run on a ventricle mesh: bench.py --mesh lv (and the shuffled box) at N > 1 partitions the cells by recursive
    coordinate bisection (tb.distributed.partition_cells_rcb / GeneralPartition: every part exchanges with every part it shares nodes with).  Two
    ranks on cuda:0 over gloo (TB_BENCH_SHARE_DEVICE) must leave behind what one rank leaves: --checksums prints partition-independent sums of
    M, K, b and the ionic states after two steps from the same initial state — equal to 1e-10 relative."""
    common = ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-graph", "--no-slab-sweep", "--checksums", "--preroll-ms", "0"] + mesh_args
    one = _bench_line(["--gpus", "1"] + common)
    two = _bench_line(["--gpus", "2"] + common, env={"TB_BENCH_SHARE_DEVICE": "1"})
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["config"]["backend"].startswith("gloo")
    assert "bisection" in two["config"]["partition"] and two["config"]["cells_total"] == one["config"]["cells_total"]
    assert two["config"]["cells_per_gpu"] * 2 in (one["config"]["cells_total"], one["config"]["cells_total"] + 1, one["config"]["cells_total"] - 1)
    assert two["phase_ms"]["halo"] > 0.0 and two["distributed_cg"]["halo_bytes_per_rank"] > 0
    for k, v in one["checksums"].items():
        assert abs(two["checksums"][k] - v) <= 1e-10 * abs(v), (k, v, two["checksums"][k])


def test_bench_three_slabs_merged_pack_and_unpack_equal_one_call_per_neighbour():
    """Three z-slabs of the box: the middle rank has two neighbours whose index lists (its bottom and top node plane) share no dof, so HaloExchange fills
    both send buffers with ONE tb_gather_indexed and adds both receive buffers with ONE tb_scatter_add_indexed (slices of one allocation each).  The run
    leaves the sums of the run that makes one call per neighbour (1e-13) (TB_HALO_SEPARATE_CALLS): the same two numbers are added on every shared dof.
    (The comparison with one rank is the next test.)"""
    common = ["--gpus", "3", "--n", "24", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-graph", "--no-slab-sweep", "--no-dist-cg", "--checksums", "--preroll-ms", "0"]
    merged = _bench_line(common, env={"TB_BENCH_SHARE_DEVICE": "1"})
    separate = _bench_line(common, env={"TB_BENCH_SHARE_DEVICE": "1", "TB_HALO_SEPARATE_CALLS": "1"})
    assert merged["n_gpus"] == 3 and merged["config"]["layers_per_gpu"] == 8 and merged["phase_ms"]["halo"] > 0.0
    for k, v in separate["checksums"].items():          # (LDS and global atomics order their adds run by run: the last bits of K and b are not reproducible)
        assert abs(merged["checksums"][k] - v) <= 1e-13 * abs(v), (k, v, merged["checksums"][k])


def test_bench_three_slabs_of_the_box_leave_the_one_rank_sums():
    """BASELINE configs[2] at N > 1 is the SAME box in N z-slabs: every rank displaces its nodes by the whole box's perturbation (global layer index), so
    the union of the slabs is the one-rank mesh, the initial state is a function of position, and ψᵀMψ, ψᵀKψ, Σ w b², Σ w bψ (b summed over the
    interfaces by the step's own exchange), Σ w u², Σ w φψ after two steps equal the one-rank run's to 1e-10 (P-GPU result == 1-GPU result)."""
    common = ["--n", "24", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-graph", "--no-slab-sweep", "--no-dist-cg", "--checksums", "--preroll-ms", "0"]
    one = _bench_line(["--gpus", "1"] + common)
    three = _bench_line(["--gpus", "3"] + common, env={"TB_BENCH_SHARE_DEVICE": "1"})
    assert three["n_gpus"] == 3 and three["config"]["layers_per_gpu"] == 8 and three["phase_ms"]["halo"] > 0.0
    for k, v in one["checksums"].items():
        assert abs(three["checksums"][k] - v) <= 1e-10 * abs(v), (k, v, three["checksums"][k])


def test_electromechanics_step_under_the_bisection_partition_matches_one_rank():
    """BASELINE configs[4] (coupled electromechanics on the ventricle, 8 GPUs) as scripts/bench_electromechanics.py --gpus N runs it: the EP operators,
    the ionic step, the distributed heat solve and the mechanics residual / tangent of every part, neighbour sums of the vectors.  Two ranks on one
    device (gloo) leave the partition-independent checksums of one rank (1e-10; the heat solution itself is iterated to rtol 1e-6 and not compared)."""
    import json
    import subprocess
    script = os.path.join(ROOT, "scripts", "bench_electromechanics.py")
    common = ["--nc", "16", "--nr", "3", "--nl", "8", "--reps", "2"]

    def line(extra, env=None):
        r = subprocess.run([sys.executable, script] + common + extra, env=dict(os.environ, **(env or {})), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])

    one = line(["--gpus", "1"])
    two = line(["--gpus", "2"], env={"TB_BENCH_SHARE_DEVICE": "1"})
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["backend"].startswith("gloo") and two["shared_nodes"] > 0
    assert two["cells_total"] == one["cells_total"] and abs(2 * two["cells_per_gpu"] - one["cells_total"]) <= 1
    for k, v in one["checksums"].items():
        assert abs(two["checksums"][k] - v) <= 1e-10 * max(abs(v), 1e-30), (k, v, two["checksums"][k])


@pytest.mark.parametrize("mode", ["all", "matrix", "none"])
def test_bench_phase_event_modes_on_one_rank(mode):
    """--phase-events: all six phase boundaries, the two around the matrix pass (what a rank of an N-rank run records in its timed steps), or none — the
    line is printed in every mode, a phase that was not recorded is NaN, the matrix phase is there unless the mode is none."""
    d = _bench_line(["--gpus", "1", "--n", "24", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-graph", "--no-slab-sweep", "--no-dist-cg", "--preroll-ms", "0",
                     "--phase-events", mode])
    ph = d["phase_ms"]
    assert d["value"] > 0 and d["host_enqueue_ms_per_step"] > 0
    assert (ph["mass+diffusion"] > 0) == (mode != "none")
    assert (ph["reaction"] > 0) == (mode == "all") and (ph["source"] > 0) == (mode == "all")
