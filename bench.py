#!/usr/bin/env python3
"""Benchmark of the hot path on N MI355X (one process per GPU).

A "step" is one pass of the hot path over one synthetic batch: the three monodomain element
integrations over the whole mesh — mass matrix M and diffusion matrix K (κ/(Cₘχ)) in one fused pass
(tb_assemble_matrix_pair: the heat stage assembles them back to back, euler.jl:172-176), source vector b —
then the halo sum of b over the slab interfaces (N > 1) and one forward-Euler reaction step of the
ionic model on every dof.

`python bench.py --gpus N` (N > 1) starts its own N ranks: the parent never touches the GPU, it runs
`python -m torch.distributed.run --nproc-per-node N … bench.py` as a child process and exits with its code.
Under torchrun (RANK / WORLD_SIZE set) WORLD_SIZE must equal --gpus.  Workload at N = 1: BASELINE.json's metric configuration, the 10M-hex Q1
monodomain mesh (216³ = 10 077 696 hexahedra, 10 218 313 dofs) with the ten Tusscher–Panfilov 2006 ionic
model that configuration names (19 states; the reference itself has no TT06 — SURVEY F6 — so its parity is
pinned against this repository's oracle only; `--ionic pcg2019` runs the reference's own 7-state model, and
its reaction rate is reported as an extra key either way).  N > 1 (default `--scaling strong`, BASELINE's configuration: the SAME 10M-hex
mesh on 1/2/4/8 GPUs): the 216³ box is cut into N z-slabs of 216/N cell layers (27 layers = 1.26 M cells per GPU at N = 8);
`--scaling weak` gives every rank a whole 216³ slab of a box N times longer in z.  Assembly has no data-path collective, the only exchange
of the step is the neighbour halo sum of shared-node vector entries; what a partitioned TIME step exchanges besides that — the interface
sums and the two all-reduces of every CG iteration of the backward-Euler solve — is timed on the same partition (barrier-bracketed, max
over ranks) and reported as a second rate, `cg_iterations_per_s`, so the line cannot scale by construction.  At N = 1 `slab_sweep` times
the step and the CG iteration on the 216×216×{108, 54, 27} slabs an N-GPU run gives each rank: t(216) / t(216/N) is the speed-up the
kernels alone allow (no exchange latency) — the evidence one GPU can give for the 1 → 8 curve.

Prints ONE JSON line (rank 0).  `value` = element-integrations/s of the whole job over the timed
region (3·cells per step / step time, reaction included in the step time); DoF-updates/s and the
per-phase rates are extra keys.
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the pool's host driver shares device memory between processes through dmabuf only (RCCL's peer buffers); exported there already, a default here
# the CPU baseline runs on the cores the container may use, pinned (docs/src/vroom.md:3-15 of the reference): libgomp reads these when it is first loaded.
# ONE-rank runs only (the baseline leg exists at N = 1 only): with the binding set, libgomp pins the INITIAL thread of every process to the first place of
# its affinity mask — under a launcher all N ranks of a node would enqueue their kernels from the same core (round 6: found by reading, no 8-GPU box here;
# at the 0.39 ms steps of the 27-layer slab eight ranks' launches on one core would sit on the clock).
_OMP_PINNED_HERE = []
if int(os.environ.get("WORLD_SIZE", "1")) <= 1:
    for _k, _v in (("OMP_PROC_BIND", "close"), ("OMP_PLACES", "cores")):
        if _k not in os.environ:
            os.environ[_k] = _v
            _OMP_PINNED_HERE.append(_k)

# The launcher sets OMP_NUM_THREADS=1 for its ranks when the caller set nothing ("to avoid your system being overloaded"): the host-side plan builders of an
# N-rank run then work on one thread each (two ranks of the 216³ box: 9.8 s of set-up).  A rank takes its share of the CPUs the container grants instead.
# Nothing inside a timed region runs on host threads.
if int(os.environ.get("WORLD_SIZE", "1")) > 1 and os.environ.get("OMP_NUM_THREADS", "1") == "1" and "TB_BENCH_KEEP_OMP" not in os.environ:
    def _cpus_granted():
        try:
            q_, p_ = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if q_ != "max":
                return max(1, int(int(q_) / int(p_)))
        except Exception:
            pass
        return os.cpu_count() or 1
    os.environ["OMP_NUM_THREADS"] = str(max(1, min(16, _cpus_granted() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))))))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)
FP64_VECTOR_TFLOPS = 78.6      # AMD spec sheet; FP64 MFMA runs at the same rate on CDNA4
BYTES_PER_CELL_MATRIX = 272.0  # 32 B conn + 24 B coords + 27 nz × 8 B
BYTES_PER_CELL_VECTOR = 64.0   # 32 + 24 + 8
BYTES_PER_DOF_UPDATE = 16.0    # in place; 24 when du is materialised
# FP64 flops of one cell instance, counted on the ISA of the shipped kernel instances (hipcc -save-temps listing of tb_patch_fused.hip; v_mul_f64 / v_add_f64 = 1,
# v_fma_f64 / v_fmac_f64 = 2), keyed by (matrices, tensor form) as tb_last_kernel_name() reports the instance that ran
FLOP_PER_INSTANCE = {("K+M", "ISO"): 375 + 354 + 2 * 483, ("K+M", "DIAG"): 430 + 354 + 2 * 465, ("K+M", "GEN"): 430 + 354 + 2 * 609,
                     ("K", "ISO"): 310 + 344 + 2 * 436, ("K", "DIAG"): 365 + 344 + 2 * 418, ("K", "GEN"): 365 + 344 + 2 * 562, ("M", "-"): 97 + 79 + 2 * 153}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=216, help="cells per edge of the box (216 → 10M hexahedra)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong = the same n³ mesh cut into N z-slabs (BASELINE's configuration); weak = one n³ slab per rank")
    ap.add_argument("--exchange", default="auto", choices=["auto", "torch", "abi"],
                    help="N > 1 data path: torch = torch.distributed over RCCL (batch_isend_irecv / all_reduce); abi = RCCL behind the C ABI (tb_comm_exchange / "
                         "tb_comm_allreduce, what a Julia host would call) in the device's own queue — torch.distributed then only carries the communicator id and "
                         "the timing barrier; auto (default) = abi where every rank can create the communicator, else torch.  Measured on one GPU with rank 0 as its own "
                         "two neighbours (profiles/r06_v1/ab_exchange_paths_of_the_n_rank_step.log, 27 layers): a step costs +0.028 ms through the C ABI, +0.075 ms "
                         "through torch.distributed (its fork to and join from c10d's queue)")
    ap.add_argument("--mesh", default="box", choices=["box", "shuffled", "lv"],
                    help="N = 1: box = the lexicographic n^3 box (BASELINE's configuration); shuffled = the SAME box with its cells and nodes randomly renumbered (same bytes, "
                         "no row / scatter signatures to share: an unstructured presentation); lv = the idealised left ventricle (O-grid apex, curved thin wall) at --lv-dims")
    ap.add_argument("--renumber", default="none", choices=["none", "dofs", "grid"],
                    help="--mesh shuffled | lv: apply tb_host_locality_permutation first — dofs = Ferrite.renumber!(dh, dof_perm) only (cells and nodes stay as "
                         "numbered), grid = the grid stored in the locality order as well (cells[cell_perm], nodes moved), then close!(dh)")
    ap.add_argument("--trace-markers", action="store_true",
                    help="launch a marker kernel (torch.cumsum on three integers: `scan` in its name, nothing else here launches one) right before and right after the timed "
                         "region, outside the clock: scripts/rocpd_summary.py --between scan then averages a rocprofv3 trace over the timed launches only")
    ap.add_argument("--checksums", action="store_true",
                    help="after the timed region: two steps from the initial state, then partition-independent sums of M, K, b and the ionic states in the line "
                         "(an N-rank run must reproduce the one-rank numbers)")
    ap.add_argument("--lv-dims", default="256,16,248", help="--mesh lv: circumferential, radial, longitudinal cell counts (default ~1.1 M hexahedra)")
    ap.add_argument("--layers", type=int, default=0, help="N = 1 diagnostics: run on the n×n×LAYERS slab only (the share of one of n/LAYERS GPUs), e.g. under rocprofv3")
    ap.add_argument("--no-spmv-mirror", action="store_true", help="CG iterations on the CSR array instead of its sliced mirror (tb_spmv_mirror)")
    ap.add_argument("--no-slab-sweep", action="store_true", help="N = 1: skip the timing of the n×n×{n/2, n/4, n/8} slabs (strong-scaling prediction)")
    ap.add_argument("--strategy", default="patch", choices=["patch", "atomic", "color"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap-exchange", action="store_true", help="N > 1: finish the halo sum of b before the reaction step starts (default: posted before, added after)")
    ap.add_argument("--preroll-ms", type=float, default=150.0,
                    help="untimed GPU work (the loop's own steps) in front of the W warm-up steps of every timed loop, so that the clock reads steady-state "
                         "throughput: after an idle period (plan building, a host-side pause) the GPU's power management takes ~10 steps of 3 ms to return to full "
                         "clocks (profiles/r05_v1/clock_ramp_trace.txt: 2.33 -> 1.65 ms for the same kernel); 0 switches it off")
    ap.add_argument("--phase-events", choices=("auto", "all", "matrix", "none"), default="auto",
                    help="HIP events recorded inside the timed steps: all six phase boundaries, only the two around the matrix pass (the roofline's kernel), or none "
                         "(diagnostic).  auto = all on one rank, matrix on N ranks, where the other phases are read from a second, instrumented pass of K steps behind "
                         "the timed region: six event records cost a 27-layer step 12 us of its 0.38 ms (profiles/r06_v1/ab_phase_event_cost.log), nothing "
                         "measurable of the whole box's 2.6 ms")
    ap.add_argument("--no-graph", action="store_true", help="N = 1: skip the HIP-graph forms of the step and of the CG iteration (tb_graph_*)")
    ap.add_argument("--no-dist-cg", action="store_true", help="skip the (untimed-region) distributed CG iteration measurement")
    ap.add_argument("--cpu-n", type=int, default=64, help="edge of the CPU-baseline sample mesh")
    ap.add_argument("--no-cpu-big", action="store_true", help="skip the one all-cores CPU pass over the GPU run's own mesh")
    ap.add_argument("--keep-du", action="store_true", help="materialise du (dumat) in the reaction step")
    ap.add_argument("--separate", action="store_true", help="assemble M and K with two launches instead of the fused pass")
    ap.add_argument("--ionic", default="tt06", choices=["pcg2019", "tt06", "fhn", "ord"],
                    help="ionic model of the reaction step (tt06 = the 19-state model BASELINE's 10M-hex configuration names; "
                         "pcg2019 = the reference's own 7-state model)")
    return ap.parse_args()


def cpu_model_string():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def physical_cores():
    """(physical cores, hardware threads) visible to this process, from /proc/cpuinfo (distinct (physical id, core id) pairs among the allowed CPUs)."""
    allowed = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
    cores, cpu, phys = set(), None, None
    try:
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "processor":
                cpu, phys = int(v), None
            elif k == "physical id":
                phys = int(v)
            elif k == "core id" and (allowed is None or cpu in allowed):
                cores.add((phys, int(v)))
    except OSError:
        pass
    nthreads = len(allowed) if allowed is not None else (os.cpu_count() or 1)
    return (len(cores) or nthreads), nthreads


def cgroup_cpu_quota():
    """CPUs' worth of CPU time the container may use (cgroup v2 cpu.max / v1 cfs quota), or None without a quota.  The GPU boxes of this pool list all
    256 hardware threads in the CPU set and cap the time at 16 CPUs: 128 pinned threads then share 16 cores' worth of time (the '7.5x from 128 cores'
    of the round-5 line), so the baseline runs as many threads as the quota grants and `cores` states that number."""
    try:
        a, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if a != "max" and float(per) > 0:
            return float(a) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and per > 0:
            return q / per
    except (OSError, ValueError):
        pass
    return None


def usable_cores():
    phys, hw = physical_cores()
    q = cgroup_cpu_quota()
    return (min(phys, max(1, int(q + 1e-9))) if q else phys), hw, phys, q


HOST_CORES = usable_cores()


def cpu_baseline(n, kap, ionic="tt06", sizes=(64, 100), big=None, check=None):
    """Oracle ("port": C restatement of the reference CPU path, NOT Julia) on bounded samples of the same workload, as SURVEY §8(d) asks: 1 thread and
    all physical cores (docs/src/vroom.md:3-15: threads = physical cores, pinned — OMP_PROC_BIND / OMP_PLACES are set at the top of this file), on the
    64³ and 100³ meshes, min over repetitions after a warm-up; the 216³ mesh of the GPU run once on all cores (`big` = its host arrays).  The
    multi-thread legs run the scaling form of the per-colour loop (oracle.AssemblyPlan: scatter positions looked up once, per-colour cell lists,
    first-touch zero fill, element routines instantiated for the hexahedron); the 1-thread legs are timed in both forms — that one and the
    literal sequential loop the parity tests use.  `value` is the all-cores rate on the n³ sample (n = --cpu-n); everything else sits in `table`."""
    from oracle import oracle as o
    cm = {"pcg2019": o.CELL_PCG2019, "tt06": o.CELL_TT06, "fhn": o.CELL_FHN, "ord": o.CELL_ORD11}[ionic]
    rdt = {"pcg2019": 0.01, "tt06": 0.001, "fhn": 0.1, "ord": 0.002}[ionic]
    p = o.cell_default_params(cm)
    cM = o.Coef(o.COEF_CONST_SCALAR, [1.0])
    cK = o.Coef(o.COEF_CONST_TENSOR, kap.ravel(), Cm=1.0, chi=1.0, wrap=True)
    cores, avail, phys_cores, quota = HOST_CORES                    # counted at import time: once libgomp has pinned the initial thread, its affinity mask is one core
    table = {}

    def legs(tag, m, rp, ci, nd, ncells, threads_list, literal_1t, median_of=0):
        col, nc = o.color_cells(m.cell_dofs, nd)
        u = np.ascontiguousarray(np.tile(o.cell_default_state(cm, p), (nd, 1)).T).ravel().copy()
        nstates = len(u) // nd
        for th in threads_list:
            plan = o.AssemblyPlan(m, rp, ci, col, nc, th)
            nzM, nzK, bvec = plan.new_values(), plan.new_values(), np.empty(nd)
            best_asm, best_rx, reps = 1e30, 1e30, 0
            asm_t, rx_t = [], []
            t_end = time.time() + (3.0 if th == 1 else 4.0)
            while reps < 2 or len(asm_t) < median_of or (time.time() < t_end and reps < 6):
                t0 = time.perf_counter()
                plan.assemble(0, cM, nzM)
                plan.assemble(1, cK, nzK)
                plan.assemble_source(o.SRC_COS_EXP, bvec, t=0.0)
                t1 = time.perf_counter()
                o.reaction_step(cm, p, u, nd, o.LAYOUT_SOA, dt=rdt, nthreads=th, want_du=False)
                t2 = time.perf_counter()
                if reps > 0 or th == 1:                            # the first multi-thread pass places the pages
                    best_asm, best_rx = min(best_asm, t1 - t0), min(best_rx, t2 - t1)
                    asm_t.append(t1 - t0); rx_t.append(t2 - t1)
                reps += 1
            ent = {"element_integrations_per_s": 3 * ncells / best_asm, "dof_updates_per_s": nstates * nd / best_rx, "reps": reps}
            if median_of:                                          # the stated baseline: median of the counted repetitions (the other entries keep @btime's minimum)
                ent.update({"element_integrations_per_s_median": 3 * ncells / float(np.median(asm_t)), "dof_updates_per_s_median": nstates * nd / float(np.median(rx_t)),
                            "counted_reps": len(asm_t)})
            table["%s/%dt" % (tag, th)] = ent
            if check is not None and median_of:                    # the oracle as the CHECKER of the GPU arrays (after every timed region): sampled rows of the bench's own mesh
                table["%s/%dt" % (tag, th)]["parity"] = check(rp, ci, nzM, nzK, bvec)
            del plan, nzM, nzK, bvec
        if literal_1t:
            t0 = time.perf_counter()
            o.assemble_matrix(m, 0, cM, rp, ci)
            o.assemble_matrix(m, 1, cK, rp, ci)
            o.assemble_source(m, o.SRC_COS_EXP, t=0.0)
            table["%s/1t-literal" % tag] = {"element_integrations_per_s": 3 * ncells / (time.perf_counter() - t0), "reps": 1,
                                            "note": "the sequential loop of coordinate_systems.jl:145-171 as the parity tests run it (generic element instances, search per entry)"}

    for nn in sorted(set(sizes) | {n}):
        xyz, conn = o.generate_grid_hex(nn, nn, nn, (0, 0, 0), (1, 1, 1))
        cd, nd = o.close_dofs(o.HEX8, 1, conn, len(xyz))
        rp, ci = o.build_pattern(cd, nd)
        legs("%d^3" % nn, o.Mesh(o.HEX8, 2, xyz, conn, cd), rp, ci, nd, nn ** 3, sorted({1, cores}), literal_1t=(nn == min(sizes)))
    if big is not None:
        xyz, conn, cd, nd, rp, ci, tag = big
        legs(tag, o.Mesh(o.HEX8, 2, xyz, conn, cd), rp, ci, nd, conn.shape[0], [cores], literal_1t=False, median_of=3)
        top = table["%s/%dt" % (tag, cores)]                       # the stated baseline: the mesh the GPU was timed on, all cores, median of >= 3 counted passes
        value, rx_value = top["element_integrations_per_s_median"], top["dof_updates_per_s_median"]
        sample = ("the bench's own %s hex Q1 mesh (%d cells): M + K per-colour + b element-assembly, OpenMP on %d pinned cores (all the container may use; OMP_PROC_BIND=%s OMP_PLACES=%s), "
                  "median of %d passes after a page-placing one; C restatement of the reference CPU path, not Julia; the 64^3 / 100^3 samples and the 1-thread legs are in `table`"
                  % (tag, conn.shape[0], cores, os.environ.get("OMP_PROC_BIND"), os.environ.get("OMP_PLACES"), top["counted_reps"]))
    else:
        top = table["%d^3/%dt" % (n, cores)]
        value, rx_value = top["element_integrations_per_s"], top["dof_updates_per_s"]
        sample = ("%d^3 hex Q1 sample mesh (%d cells; --no-cpu-big: the bench's own mesh was skipped): M + K per-colour + b element-assembly, OpenMP on %d pinned cores "
                  "(all the container may use), min of %d reps; C restatement of the reference CPU path, not Julia" % (n, n ** 3, cores, top["reps"]))
    one_t = table.get("%d^3/1t" % min(sizes), {}).get("element_integrations_per_s")
    eff = value / (cores * one_t) if one_t else None
    if eff is not None:                                            # the caveat travels with the number: what all cores deliver of `cores` × the one-thread rate
        sample += ("; measured thread efficiency %.2f (all-cores rate / (%d x the 1-thread rate of the %d^3 sample)): the per-colour loop sweeps the whole value array "
                   "once per colour and is bound by the host's memory system at this size, not by its cores" % (eff, cores, min(sizes)))
    return {"value": value, "unit": "element-integrations/s", "cores": cores, "kind": "port", "sample": sample, "thread_efficiency": eff,
            "dof_updates_per_s": rx_value, "ionic_model": ionic, **({"parity": top["parity"]} if "parity" in top else {}),
            "table": table, "physical_cores": phys_cores, "cgroup_cpu_quota": quota, "hardware_threads": avail, "cpu_model": cpu_model_string(),
            "cores_note": ("threads = min(physical cores, the container's CFS quota): the box lists %d physical cores and grants %.1f CPUs of time" % (phys_cores, quota)) if quota else None}


def spawn_ranks(args):
    """--gpus N without a launcher: start N ranks as a child torchrun (this process has not initialised the GPU and does not
    re-exec; it waits and returns the child's exit code)."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()    # counts devices without creating a context
    if have < args.gpus and not os.environ.get("TB_BENCH_SHARE_DEVICE"):
        sys.stderr.write("bench.py: --gpus %d requested but %d GPU(s) visible\n" % (args.gpus, have))
        return 2
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), "--", os.path.abspath(__file__)] + sys.argv[1:]  # "--": the launcher's argparse must not read bench.py's options (--n is a prefix of several of its own)
    env = dict(os.environ)
    for k in _OMP_PINNED_HERE:                 # the binding this process gave itself is not for its ranks (see the top of the file)
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def ferrite_pin():
    """Whether the maintainer-side golden files of julia/make_golden.jl are present: without them Ferrite's conventions (node / cell order of generate_grid,
    close!(dh) numbering, pattern order of src/solver/interface.jl:162-168, Gauss-point order) are SURVEY-§8(c) hypotheses, and every number says so."""
    gd_ = os.path.join(ROOT, "tests", "golden")
    have = sorted(f for f in (os.listdir(gd_) if os.path.isdir(gd_) else []) if f.startswith("ferrite_") and f.endswith(".json"))
    return {"ferrite_conventions": "pinned on " + ", ".join(have) if have else "UNPINNED",
            "note": None if have else "Ferrite conventions unpinned: whole-mesh K / M / b, dof numbering and pattern order are checked against the C oracle only "
                                      "(tests/golden/ferrite_*.json absent; julia/make_golden.jl writes them on a machine with Julia + Ferrite)"}


FERRITE_PIN = ferrite_pin()


def main():
    args = parse()
    if args.gpus < 1:
        sys.stderr.write("bench.py: --gpus must be >= 1\n")
        return 2
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if not launched and args.gpus > 1:
        return spawn_ranks(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write("bench.py: WORLD_SIZE=%d but --gpus %d — refusing to print a line for the wrong job size\n" % (world, args.gpus))
        return 2
    import torch
    import torch.distributed as dist
    import thunderbolt_jl_amd as tb

    # TB_BENCH_SHARE_DEVICE=1 (tests only: a one-GPU box): every rank on cuda:0 with the gloo backend, halo buffers staged through the host — the
    # same partition / exchange / timing code as the RCCL run, whose line it marks "backend": "gloo (shared device, test)"
    share = bool(os.environ.get("TB_BENCH_SHARE_DEVICE")) and world > 1
    device_index = 0 if share else local_rank
    torch.cuda.set_device(device_index)
    # under a launcher the process group is created at every world size, 1 included: the RCCL initialisation, the barrier and the max-over-ranks
    # all-reduce of the timing then run on a one-GPU box too (tests/test_rccl_world1.py), not for the first time on the 8-GPU node
    if world > 1 or launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
    dev = tb.MI355XDevice(device_index)
    # torch and the library share ONE non-default stream: ordering between the two needs no events, and the stream can be captured into HIP graphs
    # (tb_graph_*; the legacy default stream cannot)
    torch.cuda.set_stream(torch.cuda.Stream())
    dev.set_stream(torch.cuda.current_stream().cuda_stream)

    n = args.n
    if args.mesh != "box" and args.layers:
        sys.stderr.write("bench.py: --layers cuts z-slabs of the box; --mesh %s is partitioned by recursive bisection (--gpus N)\n" % args.mesh)
        return 2
    if args.mesh != "box" and world > 1 and args.scaling == "weak":
        sys.stderr.write("bench.py: --scaling weak replicates box slabs; --mesh %s runs strong scaling (one mesh, N parts)\n" % args.mesh)
        return 2
    strong = args.scaling == "strong"
    if strong and world > n:
        sys.stderr.write("bench.py: --gpus %d exceeds the %d cell layers of the mesh\n" % (world, n))
        return 2
    nz_total = n if strong else n * world
    if args.layers > 0 and world == 1:
        nz_total = args.layers
    part = tb.distributed.SlabPartition((n, n, nz_total), (0.0, 0.0, 0.0), (1.0, 1.0, nz_total / n), world, rank)
    kap = np.diag([4.5e-5, 2.0e-5, 2.0e-5])  # docs/src/literate-tutorials/ep01_spiral-wave.jl:39-41 style conductivities
    rdt = {"pcg2019": 0.01, "tt06": 0.001, "fhn": 0.1, "ord": 0.002}[args.ionic]   # forward-Euler-stable reaction step sizes
    fused = args.strategy == "patch" and not args.separate
    gd = dist if dist.is_initialized() else None
    xd = gd                                                          # what the halo exchange and the CG reductions go through
    host_red = "cpu" if share else "cuda"
    # (share = the one-GPU test configuration over gloo: RCCL refuses two ranks on one device, so the C ABI's communicator is tried there only when
    # TB_RCCL_LIBRARY names another library — tests/mock_rccl, the test double that lets the multi-rank tb_comm_* path run on one GPU)
    if gd is not None and args.exchange in ("auto", "abi") and (not share or os.environ.get("TB_RCCL_LIBRARY")):
        # every rank tries, the ranks agree (min over ranks), one path for all: a rank that cannot create the communicator must not leave the others in
        # a grouped send / receive nobody answers
        xa_, why_ = None, ""
        try:
            if os.environ.get("TB_BENCH_NO_ABI_COMM"):               # (tests: the fall-back branch below)
                raise RuntimeError("TB_BENCH_NO_ABI_COMM is set")
            xa_ = tb.distributed.RcclComm.from_torch(dev, dist)
        except Exception as ex:
            why_ = str(ex)[:300]
        ok_ = torch.tensor([1 if xa_ is not None else 0], dtype=torch.int32, device=host_red)
        dist.all_reduce(ok_, op=dist.ReduceOp.MIN)
        if int(ok_.item()) == 1:
            xd = xa_
            if rank == 0 and world > 1:
                sys.stderr.write("bench.py: halo exchange and CG reductions through the C ABI's RCCL communicator (tb_comm_*), in the device's queue\n")
        else:
            if xa_ is not None:
                xa_.close()
            if args.exchange == "abi":
                sys.stderr.write("bench.py: --exchange abi: the communicator of the C ABI could not be created on every rank (%s)\n" % (why_ or "another rank failed"))
                return 2
            if rank == 0:
                sys.stderr.write("bench.py: --exchange auto: no communicator behind the C ABI (%s): torch.distributed carries the exchange\n" % (why_ or "another rank failed"))

    # unstructured presentations: every rank builds the whole mesh (host-side generators), bisects the cell centroids into `world` parts of equal size
    # (tb.distributed.partition_cells_rcb — the partition of SURVEY §8e for meshes that are not boxes) and keeps its own part as a local grid with
    # the nodes it shares with every other part (GeneralPartition: both sides list them in ascending global node id, so the buffers line up)
    full_grid, gpart = None, None
    if args.mesh != "box":
        if args.mesh == "lv":
            nc_, nr_, nl_ = (int(v) for v in args.lv_dims.split(","))
            full_grid = tb.generate_ideal_lv_mesh_hex(nc_, nr_, nl_)
        else:                                  # the same cells and nodes under random numbers: no lexicographic regularity for the plans to share (seeded)
            g0_ = tb.generate_mesh(tb.Hexahedron, (n, n, nz_total), (0.0, 0.0, 0.0), (1.0, 1.0, nz_total / n), perturb=0.2)
            rng = np.random.default_rng(2025)
            pn, pc = rng.permutation(g0_.n_nodes), rng.permutation(g0_.n_cells)
            inv = np.empty_like(pn); inv[pn] = np.arange(g0_.n_nodes)
            full_grid = tb.Grid(tb.Hexahedron, g0_.xyz[pn], inv[g0_.conn[pc]].astype(np.int32))
            del g0_, pn, pc, inv
        if world > 1:
            cen_ = full_grid.xyz[full_grid.conn].mean(axis=1)
            gpart = tb.distributed.GeneralPartition(full_grid.conn, tb.distributed.partition_cells_rcb(cen_, world), rank)
            del cen_

    class Problem:
        """One rank's share of the workload: mesh slab, operators, source vector, halo exchange, ionic states; `step` is one pass of the hot path."""

        def __init__(self, nel, left, right, rank_, world_, lo_up, dist_, xdist_=None):
            self.world, self.rank, self.dist = world_, rank_, dist_     # dist: barrier / max over ranks of the timings (torch.distributed)
            self.xdist = xdist_ if xdist_ is not None else dist_        # xdist: halo exchange and CG reductions (torch.distributed or RcclComm)
            if args.mesh == "box" and world_ > 1:
                # the slab of the WHOLE box's perturbed mesh: tb_host_perturb_nodes displaces the nodes of the lattice it is given and keeps that lattice's
                # boundary planes, so a slab perturbed on its own is a conforming piece of another mesh.  The same displacement by GLOBAL layer index here:
                # the union of the N slabs is the one-rank mesh (tests: N ranks leave the one-rank sums), interface planes included
                self.g = g = tb.generate_mesh(tb.Hexahedron, nel, left, right, perturb=0.0)
                px_, py_, pz_ = nel[0] + 1, nel[1] + 1, nel[2] + 1
                X_ = g.xyz.reshape(pz_, py_, px_, 3)
                h_ = [(X_[-1, -1, -1, d_] - X_[0, 0, 0, d_]) / nel[d_] for d_ in range(3)]
                twopi_ = 6.283185307179586
                s_ = (np.sin(twopi_ * (np.arange(pz_) + part.z0) / nz_total)[:, None, None] * np.sin(twopi_ * np.arange(py_) / nel[1])[None, :, None]
                      * np.sin(twopi_ * np.arange(px_) / nel[0])[None, None, :])
                X_[..., 0] += 0.2 * h_[0] * s_
                X_[..., 1] -= 0.5 * 0.2 * h_[1] * s_
                X_[..., 2] += 0.75 * 0.2 * h_[2] * s_
            elif args.mesh == "box":
                self.g = g = tb.generate_mesh(tb.Hexahedron, nel, left, right, perturb=0.2)
            elif gpart is not None:            # this rank's part of the bisected mesh, nodes numbered by first visit
                self.g = g = tb.Grid(full_grid.cell_kind, full_grid.xyz[gpart.global_nodes], gpart.conn)
            else:
                self.g = g = full_grid
            self.node_perm = None              # --renumber grid: new number of every node of g as it was handed in (shared-node lists follow)
            self.renumber_s = 0.0
            if args.renumber == "grid":
                t0_ = time.perf_counter()
                cp_, np_, _ = tb.locality_permutation(g)
                self.g = g = tb.renumber_grid(g, cp_, np_)
                self.node_perm = np_
                self.renumber_s = time.perf_counter() - t0_
            self.dh = dh = tb.DofHandler(g)
            if args.renumber == "dofs":
                t0_ = time.perf_counter()
                _, _, dp_ = tb.locality_permutation(g, dh)
                self.dh = dh = tb.renumber_dofs(dh, dp_)
                self.renumber_s = time.perf_counter() - t0_
            self.sp = sp = tb.allocate_matrix(dh)
            st = {"patch": tb.PatchAssemblyStrategy, "atomic": tb.AtomicAssemblyStrategy, "color": tb.PerColorAssemblyStrategy}[args.strategy](dev)
            # linear form: the atomic strategy on hexahedra is the patch-reduced kernel k_vector_hex8_patch<false> — a patch integrates its own cells only
            # (no halo cells to re-integrate), sums them per node in LDS and adds every touched node to the zeroed vector with one global atomic
            # (1.4 atomics per cell); the halo flavour (PATCH) stores each dof once but re-integrates 49 % more cells: 0.53 vs 0.34 ms at 216³
            st_vec = tb.AtomicAssemblyStrategy(dev) if args.strategy == "patch" else st
            D = tb.ConductivityToDiffusivityCoefficient(tb.ConstantCoefficient(kap), tb.ConstantCoefficient(1.0), tb.ConstantCoefficient(1.0))
            self.M = tb.setup_operator(st, tb.BilinearMassIntegrator(tb.ConstantCoefficient(1.0)), dh, sp)
            self.K = tb.setup_operator(st, tb.BilinearDiffusionIntegrator(D), dh, sp)
            self.src = tb.setup_operator(st_vec, tb.LinearIntegrator(tb.AnalyticalCoefficient("cos_exp")), dh)
            # source vector lives in a torch tensor so RCCL can exchange its interface entries
            self.b = torch.zeros(dh.ndofs, dtype=torch.float64, device="cuda")
            self.src.b = tb.DeviceVector.wrap(dev, self.b)
            self.lo_idx = self.up_idx = None
            nbrs = []
            if world_ > 1:
                n2d = tb.distributed.node_to_dof(dh)
                nmap = (lambda v: v) if self.node_perm is None else (lambda v: self.node_perm[v])
                if gpart is not None:          # general partition: one index list per neighbouring part (local nodes → dofs; peers ascending on both sides)
                    nbrs = [(peer, torch.from_numpy(n2d[nmap(idx)]).cuda()) for peer, idx in gpart.neighbours]
                else:
                    lo, up = lo_up
                    self.lo_idx = None if lo is None else torch.from_numpy(n2d[nmap(lo)]).cuda()
                    self.up_idx = None if up is None else torch.from_numpy(n2d[nmap(up)]).cuda()
                    nbrs = tb.distributed.slab_neighbours(self.lo_idx, self.up_idx, rank_, world_)
            self.nbrs = nbrs
            # the one data-path exchange: persistent send / receive buffers, pack and unpack through the C ABI (tb_gather_indexed / tb_scatter_add_indexed)
            self.halo = tb.distributed.HaloExchange(nbrs, self.xdist, self.b, dev)
            if os.environ.get("TB_BENCH_HALO_OWN_QUEUE") and isinstance(self.xdist, tb.distributed.RcclComm):
                self.halo.overlap = True           # (tests, comparisons) tb_comm_exchange_begin / _end: the exchange on the communicator's own queue
            # TB_BENCH_SELF_EXCHANGE=torch|abi|abi-overlap (one rank, a diagnostic): the N-rank step as it is — pack, exchange, unpack-add through HaloExchange, rank 0
            # as its own lower and upper neighbour (bottom and top node plane) — over torch.distributed (RCCL at world size 1), or the C ABI's communicator
            # in the device's queue / on its own queue.  b is summed with itself on the two planes: timings only.
            self.selfhalo = False
            sh_ = os.environ.get("TB_BENCH_SELF_EXCHANGE", "") if world_ == 1 and args.mesh == "box" else ""
            if sh_ in ("torch", "abi", "abi-overlap"):
                n2d = tb.distributed.node_to_dof(dh)
                npl_ = (nel[0] + 1) * (nel[1] + 1)
                planes = [torch.from_numpy(n2d[np.arange(npl_)]).cuda(), torch.from_numpy(n2d[np.arange(g.n_nodes - npl_, g.n_nodes)]).cuda()]
                if sh_ == "torch":
                    import torch.distributed as d1_
                    if not d1_.is_initialized():
                        d1_.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % (29500 + os.getpid() % 2000), rank=0, world_size=1)
                    xd1_ = d1_
                else:
                    xd1_ = tb.distributed.RcclComm(dev, 0, 1)
                self.halo = tb.distributed.HaloExchange([(0, planes[0]), (0, planes[1])], xd1_, self.b, dev)
                self.halo.overlap = sh_ == "abi-overlap"
                self.selfhalo = True
            self.model = model = {"pcg2019": tb.PCG2019, "tt06": tb.TT06, "fhn": tb.FHNModel, "ord": tb.ORd2011}[args.ionic]()
            self.ns = model.nstates
            self.npts = npts = dh.ndofs
            u0 = np.tile(model.default_initial_state(), (npts, 1))
            if args.mesh == "box":
                # a ramp along z by the dof's POSITION (through round 5: by its index, which on a slab is another state than on the whole box): V rises from
                # rest to rest + 60 mV over the box, the same physical state under every partition (--checksums)
                zd_ = np.empty(npts); zd_[tb.distributed.node_to_dof(dh)] = g.xyz[:, 2]
                zlo_, zhi_ = (0.0, nz_total / n) if world_ > 1 else (left[2], right[2])
                u0[:, model.phi_index] += (60.0 if self.ns > 2 else 1.0) * (zd_ - zlo_) / (zhi_ - zlo_)
            else:                                  # a function of the dof's position: the same physical state under every numbering and every partition (--checksums)
                xd_ = np.empty((npts, 3)); xd_[tb.distributed.node_to_dof(dh)] = g.xyz
                self.xdof = xd_
                u0[:, model.phi_index] += (30.0 if self.ns > 2 else 0.5) * (1.0 + np.sin(3.0 * xd_[:, 0] + 2.0 * xd_[:, 1] + xd_[:, 2]))
            self.u0 = np.ascontiguousarray(u0.T).ravel()
            self.u = torch.from_numpy(self.u0).cuda()
            self.f = tb.PointwiseODEFunction(npts, model)
            self.cache = tb.setup_solver_cache(self.f, tb.ForwardEulerCellSolver(dev), u=self.u, keep_du=args.keep_du)
            self.evpool = []
            # phase → accumulated ms; the key order is the launch order of `step` (fused: "diffusion" holds the M + K pass, "mass" stays 0)
            # TB_BENCH_SELF_EXCHANGE=first|late (one rank, a diagnostic): a real RCCL exchange in the step — rank 0 as its own two neighbours, one interface
            # plane each way on the communicator's queue — posted where an N-rank step posts it (first) or in front of the reaction kernel (late)
            self.selfx = None
            sx_ = os.environ.get("TB_BENCH_SELF_EXCHANGE", "") if world_ == 1 else ""
            if sx_ in ("first", "late"):
                npl_ = (nel[0] + 1) * (nel[1] + 1)
                self.selfx = (tb.distributed.RcclComm(dev, 0, 1), [torch.zeros(npl_, dtype=torch.float64, device="cuda") for _ in range(2)],
                              [torch.empty(npl_, dtype=torch.float64, device="cuda") for _ in range(2)])
            self.pe_mode = args.phase_events if args.phase_events != "auto" else ("all" if world_ == 1 else "matrix")
            one_rank_order = world_ == 1 and sx_ != "first" and not self.selfhalo
            order = ("mass", "diffusion", "source", "halo", "reaction") if one_rank_order else ("source", "halo", "mass", "diffusion", "reaction")
            self.phase = {k_: 0.0 for k_ in order}

        def step(self, i, ev=None):
            """one step; ev: six HIP events of THIS step (phase boundaries), read only after the timed region's synchronize — reading them here
            would wait for the GPU every step, and the host's enqueue time of the next step would sit on the clock (it did through round 4:
            the same step replayed as a graph was 4–6 % faster, most of it this wait).
            The three forms are independent, so the order of their launches is free.  One rank: M + K, b, reaction.  N > 1: b FIRST, its interface
            entries packed and the exchange posted, then M + K and the reaction step, then the received partials added.  Through the C ABI (--exchange
            auto) the grouped send / receive sits in the device's queue right behind the pack; where the exchange runs on a queue of its own
            (torch.distributed, tb_comm_exchange_begin) its kernel needs a few workgroups of this GPU, and beside the fused matrix pass (thousands of
            11 µs workgroups) it finds them at once, where the reaction kernel of a thin slab is ONE generation of resident workgroups that hold every
            register file until the kernel ends (--no-overlap-exchange: b, exchange, M + K, reaction).  "halo" = pack + post (+ the whole exchange
            when it is in the queue); "reaction" then includes the wait and the unpack."""
            t = 0.01 * i
            allev = ev is not None and self.pe_mode == "all"
            mev = ev is not None and self.pe_mode != "none"

            def mass():
                if not fused:
                    tb.update_operator(self.M, t)

            def matrix():
                if fused:
                    tb.update_operators(self.M, self.K, t)
                else:
                    tb.update_operator(self.K, t)
                if i == 0:
                    self.matrix_kernel = tb.lib().tb_last_kernel_name().decode()   # the instance this call launched (the roofline's kernel)

            def source():
                tb.update_operator(self.src, t)

            def halo():
                if self.selfx:
                    self.selfx[0].exchange([0, 0], self.selfx[1], self.selfx[2], overlapped=True)
                if self.world > 1 or self.selfhalo:
                    if args.no_overlap_exchange:
                        self.halo.exchange_sum(self.b)
                    else:
                        self.halo.pack(self.b); self.halo.start()

            def reaction():
                tb.perform_step(self.f, self.cache, t, rdt)
                if self.selfx:
                    self.selfx[0].exchange_end()
                if (self.world > 1 or self.selfhalo) and not args.no_overlap_exchange:
                    self.halo.finish(self.b)

            run = {"mass": mass, "diffusion": matrix, "source": source, "halo": halo, "reaction": reaction}
            names = list(self.phase)                                   # the order of the launches = the order of the keys (set in __init__ by the world size)
            is_matrix = [n == "diffusion" or (n == "mass" and not fused) for n in names]
            for k, name in enumerate(names):
                if allev or (mev and (is_matrix[k] or (k > 0 and is_matrix[k - 1]))): ev[k].record()
                run[name]()
            if allev or (mev and is_matrix[-1]): ev[len(names)].record()

        def step_plain(self, t):
            """the calls of `step` without the phase events (what a captured step replays; one rank, no halo exchange)"""
            if fused:
                tb.update_operators(self.M, self.K, t)
            else:
                tb.update_operator(self.M, t)
                tb.update_operator(self.K, t)
            tb.update_operator(self.src, t)
            tb.perform_step(self.f, self.cache, t, rdt)

        def checksums(self):
            """Partition- and numbering-independent sums of what two steps from the initial state leave behind (--checksums): every rank adds its own
            share — shared dofs weighted 1 / multiplicity, sub-domain operators through their quadratic forms ψᵀA_pψ with ψ a function of the dof's
            position — and the ranks' shares are all-reduced.  An N-rank run must print the numbers of the one-rank run (tests: 1e-10 relative)."""
            npts = self.npts
            self.u.copy_(torch.from_numpy(self.u0))
            for i in range(2):
                self.step(i)
            torch.cuda.synchronize()
            w = torch.ones(npts, dtype=torch.float64, device="cuda")
            for _, idx in self.nbrs:
                w[idx] += 1.0
            w = 1.0 / w
            xd_ = getattr(self, "xdof", None)
            if xd_ is None:
                xd_ = np.empty((npts, 3)); xd_[tb.distributed.node_to_dof(self.dh)] = self.g.xyz
            psi = torch.from_numpy(np.cos(2.0 * xd_[:, 0] - xd_[:, 1]) + 0.5 * np.sin(4.0 * xd_[:, 2]) + 1.5).cuda()
            y = torch.empty_like(psi)
            sums = []
            for op_ in (self.M, self.K):
                op_.mul(tb.DeviceVector.wrap(dev, y), tb.DeviceVector.wrap(dev, psi))
                sums.append((psi * y).sum())
            sums += [(w * self.b * self.b).sum(), (w * self.b * psi).sum()]
            U = self.u.view(self.ns, npts)
            sums += [(w * U * U).sum(), (w * U[self.model.phi_index] * psi).sum()]
            t_ = torch.stack(sums)
            if self.world > 1:
                tb.distributed.all_reduce_sum(t_, self.xdist)
            names = ("psi_M_psi", "psi_K_psi", "b_b", "b_psi", "u_u", "phi_psi")
            return {k_: float(v_) for k_, v_ in zip(names, t_.cpu().numpy())}

        def time_steps_graph(self, warmup, steps):
            """The same step as ONE HIP-graph launch (tb_graph_*: the sequence captured once, the time of every replay set through the device's time
            slot): W untimed launches, K timed ones, synchronize on both sides.  One rank only (the halo exchange stays outside graphs)."""
            assert self.world == 1
            gc.collect(); gc.disable()                               # (a collector pass over the set-up's arrays is a 40–80 ms pause)
            try:                                                     # (a failed capture must not leave the collector off for the measurements behind it)
                gr = dev.capture(lambda: self.step_plain(0.0))
                self.preroll(lambda i: gr.launch(0.01 * i))
                for i in range(warmup):
                    gr.launch(0.01 * i)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(steps):
                    gr.launch(0.01 * (warmup + i))
                torch.cuda.synchronize()
                el = time.perf_counter() - t0
            finally:
                gc.enable()
            dev.poll_status()
            nodes = gr.nodes
            gr.close()
            return el, nodes

        def sync(self):
            if self.dist is not None:
                self.dist.barrier()
            torch.cuda.synchronize()

        def preroll(self, fn):
            """args.preroll_ms of untimed launches of fn(i) back to back, so that a timed loop starts on a GPU at its steady clocks"""
            if args.preroll_ms <= 0:
                return
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(3):
                fn(i)
            torch.cuda.synchronize()
            per = self.max_over_ranks(max((time.perf_counter() - t0) / 3, 1e-5))   # one figure for all ranks: fn may hold an exchange, every rank must launch it equally often
            n = int(min(max(args.preroll_ms * 1e-3 / per, 1), 2000))
            for i in range(n):
                fn(i)
            torch.cuda.synchronize()

        def max_over_ranks(self, x):
            if self.dist is not None:
                tt = torch.tensor([x], dtype=torch.float64, device=host_red)
                self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
                x = float(tt.item())
            return x

        def time_steps(self, warmup, steps):
            """W untimed steps, then exactly K steps bracketed by barrier + synchronize on both sides; the MAX over ranks of the wall time"""
            while len(self.evpool) < steps:                          # six events per timed step, created outside the clock
                self.evpool.append([dev.event() for _ in range(6)])
            gc.collect(); gc.disable()                               # no collector pause (freeing buffers of earlier set-up) inside the region — nor in front of it, where it would let the GPU idle
            self.step(0)                                             # status read after every assembly call: a bad mesh or pattern surfaces here
            dev.defer_status(True)
            self.preroll(lambda i: self.step(i))                     # untimed: clocks (see --preroll-ms)
            dev.defer_status(False)
            for i in range(warmup):
                self.step(i)                                         # the W warm-up steps, status read per call
            if args.trace_markers:
                self.marker = torch.arange(3, device="cuda")
                torch.cumsum(self.marker, 0)
            self.sync()
            # the timed steps run with the status deferred (tb_device_defer_status): the mesh is fixed, so the per-call check of the warm-up steps
            # is what a time loop needs; the steps enqueue back to back and the one status read of the region sits inside it, before the clock stops
            dev.defer_status(True)
            for k_ in self.phase:
                self.phase[k_] = 0.0
            t0 = time.perf_counter()
            for i in range(steps):
                self.step(warmup + i, self.evpool[i])
            self.host_enqueue_ms = (time.perf_counter() - t0) / steps * 1e3    # what the host needs to enqueue a step (≥ ms_per_step: the host is the limiter)
            torch.cuda.synchronize()
            gc.enable()
            dev.poll_status()
            if self.dist is not None:
                self.dist.barrier()
            dt_ = self.max_over_ranks(time.perf_counter() - t0)
            if args.trace_markers:
                torch.cumsum(self.marker, 0)
                torch.cuda.synchronize()
            dev.defer_status(False)
            for i in range(steps):                                   # phase durations of the timed steps (the events are complete: no wait)
                for k, name in enumerate(self.phase):
                    recorded = self.pe_mode == "all" or (self.pe_mode == "matrix" and (name == "diffusion" or (name == "mass" and not fused)))
                    self.phase[name] += self.evpool[i][k].elapsed_ms(self.evpool[i][k + 1]) if recorded else float("nan")
            return dt_

        def instrumented_phases(self, steps):
            """K more steps with all six phase events, outside the timed region (every rank: a step holds an exchange): the phases the timed steps did not
            record (--phase-events auto on N ranks).  Returns {phase: summed ms}; the matrix phase of the line stays the timed region's."""
            keep, self.pe_mode = self.pe_mode, "all"
            try:
                while len(self.evpool) < steps:
                    self.evpool.append([dev.event() for _ in range(6)])
                dev.defer_status(True)
                for i in range(3):
                    self.step(i)
                self.sync()
                for i in range(steps):
                    self.step(i, self.evpool[i])
                torch.cuda.synchronize()
                dev.defer_status(False)
                dev.poll_status()
                out_ = {name: 0.0 for name in self.phase}
                for i in range(steps):
                    for k, name in enumerate(self.phase):
                        out_[name] += self.evpool[i][k].elapsed_ms(self.evpool[i][k + 1])
            finally:
                self.pe_mode = keep
            return out_

        def time_cg(self, nit=10):
            """What a time step adds to the assembly under a partition — one iteration of the distributed Jacobi-CG on the heat matrix A = M − Δt·K of
            this rank's slab: interface rows of A·p packed first (tb_spmv_csr_rows), exchange posted, whole local product + pᵀA_p p behind it
            (tb_spmv_csr_dot), received partials added, all-reduce of pᵀAp, update with the two weighted sums, all-reduce, direction.  Barrier-bracketed,
            max over ranks, device scalars, no host read."""
            npts, world_, dist_, xd_ = self.npts, self.world, self.dist, self.xdist
            gc.collect(); gc.disable()                               # no collector pause inside the short timed loops below
            A = tb.heat_system_matrix(dev, self.M, self.K, 0.01)
            diag = torch.empty(npts, dtype=torch.float64, device="cuda")
            tb._lib.check(tb.lib().tb_extract_diagonal(self.K.pattern.h, A.ptr, diag.data_ptr()))
            cg = tb.distributed.DistributedCG(None, diag, self.lo_idx, self.up_idx, self.rank, world_, xd_, neighbours=self.nbrs, device=dev, operator=(self.K.pattern, A))
            # the solve multiplies one fixed matrix: its values are mirrored slice by slice once (tb_spmv_mirror, timed on its own) and every product of
            # the iterations below streams the mirror; --no-spmv-mirror times the CSR kernel instead
            e0, e1 = dev.event(), dev.event()
            mirrored = (not args.no_spmv_mirror) and self.K.pattern.mirror(A)    # first bind of the pattern: slice table (host) + buffer
            e0.record()
            if mirrored:
                self.K.pattern.mirror(A)                                         # what every later matrix of this pattern costs
            e1.record()
            torch.cuda.synchronize()
            bind_ms = e0.elapsed_ms(e1) if mirrored else None
            xs_ = torch.zeros(npts, dtype=torch.float64, device="cuda")
            rs_ = self.b.clone() + 1.0
            ps_ = cg.dinv * rs_
            Ap_ = torch.empty_like(xs_)
            S_ = torch.zeros(6, dtype=torch.float64, device="cuda")
            tb._lib.check(tb.lib().tb_cgd_dot(dev.h, npts, cg.w.data_ptr(), rs_.data_ptr(), ps_.data_ptr(), S_[0:1].data_ptr()))
            if world_ > 1:
                tb.distributed.all_reduce_sum(S_[0:1], xd_)
            self.preroll(lambda i: cg.device_step(xs_, rs_, ps_, Ap_, S_))
            for _ in range(3):
                cg.device_step(xs_, rs_, ps_, Ap_, S_)
            self.sync()
            ea, eb, ec = dev.event(), dev.event(), dev.event()
            t0 = time.perf_counter()
            for _ in range(nit):
                cg.device_step(xs_, rs_, ps_, Ap_, S_)
            torch.cuda.synchronize()
            if dist_ is not None:
                dist_.barrier()
            t_it = self.max_over_ranks((time.perf_counter() - t0) / nit)
            graph_it = graph_nodes = graph_err = None
            if world_ == 1 and not args.no_graph:                     # the same iteration as one HIP-graph launch (no exchange at one rank)
                try:
                    gr = dev.capture(lambda: cg.device_step(xs_, rs_, ps_, Ap_, S_))
                    self.preroll(lambda i: gr.launch(0.0))
                    for _ in range(3):
                        gr.launch(0.0)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(nit):
                        gr.launch(0.0)
                    torch.cuda.synchronize()
                    graph_it, graph_nodes = (time.perf_counter() - t0) / nit, gr.nodes
                    gr.close()
                except Exception as ex:                               # an unsupported graph call must not cost the line
                    graph_err = str(ex)[:200]
            ea.record()
            for _ in range(nit):
                cg.device_iteration(ps_, Ap_, S_)                    # product + halo + pᵀAp only
            eb.record()
            for _ in range(nit):
                tb._lib.check(tb.lib().tb_spmv_csr(self.K.pattern.h, A.ptr, ps_.data_ptr(), 1.0, 0.0, Ap_.data_ptr()))   # the bare local SpMV, for reference
            ec.record()
            self.K.pattern.mirror(None)
            for _ in range(nit):
                tb._lib.check(tb.lib().tb_spmv_csr(self.K.pattern.h, A.ptr, ps_.data_ptr(), 1.0, 0.0, Ap_.data_ptr()))   # … and from the CSR array
            ed = dev.event()
            ed.record()
            torch.cuda.synchronize()
            gc.enable()
            return {"iteration_ms": t_it * 1e3, "graph_iteration_ms": None if graph_it is None else graph_it * 1e3, "graph_nodes": graph_nodes, "graph_error": graph_err,
                    "product_halo_dot_ms": ea.elapsed_ms(eb) / nit, "local_spmv_ms": eb.elapsed_ms(ec) / nit,
                    "local_spmv_csr_ms": ec.elapsed_ms(ed) / nit, "spmv_mirror": bool(mirrored), "spmv_mirror_bind_ms": bind_ms,
                    # algorithmic bytes of one iteration: the matrix values once (8 B per non-zero) + 12–13 passes over vectors of the rows (product:
                    # p in, Ap out; update: p, Ap, x, r, D⁻¹, weights in, x, r out; direction: r, D⁻¹, p in, p out) — against the 8 TB/s of the guide
                    "bytes_per_iteration": 8.0 * self.sp.nnz + (13 if cg.nb else 12) * 8.0 * npts,     # (no weights read without shared dofs)
                    "hbm_frac": (8.0 * self.sp.nnz + (13 if cg.nb else 12) * 8.0 * npts) / (t_it * 8.0e12),
                    "halo_bytes_per_rank": self.halo.nbytes, "rows_per_rank": npts, "nnz_per_rank": self.sp.nnz,
                    "note": "one Jacobi-CG iteration on A = M - dt K under the partition (barrier-bracketed, max over ranks, host-timed over %d iterations, device "
                            "scalars, no host read): interface rows packed first, exchange overlapped with the local SpMV + p'Ap, two all-reduces" % nit}

    t_setup0 = time.perf_counter()
    pr = Problem(part.local_nel(), part.left, part.right, rank, world, part.interface_nodes(), gd, xd)
    g, dh, sp, npts, ns = pr.g, pr.dh, pr.sp, pr.npts, pr.ns
    t_setup1 = time.perf_counter()
    if world > 1:
        # the exchange carries what it should before anything is timed through it: every rank sends rank + 1 on its shared dofs and must hold
        # rank + 1 + Σ (peer + 1) over the parts it shares each dof with — exactly (small integers), on whichever path was chosen above
        v_ = torch.full((npts,), float(rank + 1), dtype=torch.float64, device="cuda")
        e_ = v_.clone()
        for peer_, idx_ in pr.nbrs:
            e_[idx_] += float(peer_ + 1)
        pr.halo.exchange_sum(v_)
        torch.cuda.synchronize()
        if not torch.equal(v_, e_):
            raise RuntimeError("bench.py: rank %d: the halo exchange returned wrong sums on %d of its dofs" % (rank, int((v_ != e_).sum())))
        del v_, e_
    pr.step(0)                                   # first assembly: the plans (patches, signatures, one-trip records) are built here
    torch.cuda.synchronize()
    t_setup2 = time.perf_counter()
    elapsed = pr.time_steps(args.warmup, args.steps)
    phase = pr.phase
    phase_note = None
    if pr.pe_mode == "matrix" and args.phase_events == "auto":
        inst_ = pr.instrumented_phases(args.steps)
        for k_ in phase:
            if phase[k_] != phase[k_]:                    # not recorded in the timed steps
                phase[k_] = inst_[k_]
        phase_note = ("the timed steps record the two events around the matrix pass only; source / halo / reaction are from a second pass of %d steps with all six "
                      "events, behind the timed region (its matrix phase: %.4f ms)" % (args.steps, inst_["diffusion"] / args.steps))

    def graph_step_ms(prob):
        """(ms per step, nodes, error) of the step as one HIP-graph launch; never fatal"""
        if world != 1 or args.no_graph:
            return None, None, None
        try:
            el_, nodes_ = prob.time_steps_graph(max(2, args.warmup), max(5, args.steps))
            return el_ / max(5, args.steps) * 1e3, nodes_, None
        except Exception as ex:
            try:
                dev.defer_status(False)
            except Exception:
                pass
            return None, None, str(ex)[:200]

    g_ms, g_nodes, g_err = graph_step_ms(pr)
    chk = pr.checksums() if args.checksums else None

    # outside the timed region: the reference's own 7-state ionic model (PCG2019) on the same points, for the record
    ref_rx = None
    if args.ionic != "pcg2019":
        mref = tb.PCG2019()
        fref = tb.PointwiseODEFunction(npts, mref)
        uref = torch.from_numpy(np.ascontiguousarray(np.tile(mref.default_initial_state(), (npts, 1)).T).ravel()).cuda()
        cref = tb.setup_solver_cache(fref, tb.ForwardEulerCellSolver(dev), u=uref, keep_du=False)
        tb.perform_step(fref, cref, 0.0, 0.01)
        ea, eb = dev.event(), dev.event()
        ea.record()
        for i in range(5):
            tb.perform_step(fref, cref, 0.01 * (i + 1), 0.01)
        eb.record()
        torch.cuda.synchronize()
        ms_ref = ea.elapsed_ms(eb) / 5
        ref_rx = {"model": "pcg2019", "states": 7, "ms": ms_ref, "dof_updates_per_s": 7 * npts / (ms_ref * 1e-3),
                  "hbm_frac": BYTES_PER_DOF_UPDATE * 7 * npts / (ms_ref * 1e-3) / 1e9 / HBM_PEAK_GBS}
        del cref, uref

    # the second rate of the line: CG iterations of the heat solve on the same partition (its own timed region, barrier-bracketed, max over ranks)
    dist_cg = None if args.no_dist_cg else pr.time_cg()

    # N = 1: the slabs a strong-scaling run on N GPUs hands each rank, timed one after the other on this GPU (same kernels, same plans, no exchange)
    slab_sweep = None
    if world == 1 and not args.no_slab_sweep and n % 8 == 0 and args.layers == 0 and args.mesh == "box":
        slab_sweep = {"layers": {}, "note": "step = M + K + b + %s reaction, cg = one Jacobi-CG iteration on A = M - dt K, each on an %dx%dxL slab of the mesh (the share of one of "
                                           "N = %d/L GPUs under --scaling strong); predicted_speedup = t(%d layers) / t(L layers): what the kernels alone allow, without the halo "
                                           "exchange and the two all-reduces of an iteration; *_graph: the same work as one HIP-graph launch per step / iteration (tb_graph_*); "
                                           "*_with_exchange: best of the two forms on both sides, plus the measured world-size-1 RCCL cost of the step's halo sum / of an "
                                           "iteration's two all-reduces (exchange_latency); the slabs are timed as a rank of an N-rank run times its steps (--phase-events auto: the two events "
                                           "around the matrix pass; phase_ms from an instrumented pass behind it)" % (args.ionic.upper(), n, n, n, n)}
        base = {"step_ms": elapsed / args.steps * 1e3, "cg_iteration_ms": dist_cg["iteration_ms"] if dist_cg else None,
                "graph_step_ms": g_ms, "graph_cg_iteration_ms": dist_cg["graph_iteration_ms"] if dist_cg else None}
        # what an exchange costs on this box: RCCL at world size 1 behind the C ABI, rank 0 as its own lower and upper neighbour (one interface plane each
        # way: (n+1)² doubles) and the 8-byte all-reduce of a CG iteration — no xGMI hop in it, but the launch, the protocol and the copy kernels
        xch = None
        try:                                                         # in a child process: a crash or a hang of the communication library must not cost the line
            import subprocess
            r_ = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "rccl_latency.py"), str((n + 1) * (n + 1))], capture_output=True, text=True, timeout=180)
            js_ = [l_ for l_ in r_.stdout.splitlines() if l_.startswith("{")]   # (RCCL prints its version banner behind the line at exit)
            xch = json.loads(js_[-1]) if r_.returncode == 0 and js_ else {"error": "rc %d: %s" % (r_.returncode, r_.stderr[-300:])}
        except Exception as ex:
            xch = {"error": str(ex)[:200]}
        slab_sweep["exchange_latency"] = xch
        slab_sweep["layers"][str(n)] = dict(base, cells=g.n_cells)
        for N in (2, 4, 8):
            L_ = n // N
            q = Problem((n, n, L_), (0.0, 0.0, 0.0), (1.0, 1.0, L_ / n), 0, 1, (None, None), None)
            # the better of two timings (this is a prediction from kernel times, not the contract's timed region: a stall of the host — the
            # collector freeing the previous slab's multi-GB buffers, a page-in — once put 20 ms into a 0.9 ms step)
            gc.collect(); torch.cuda.synchronize()
            # timed as a rank of an N-rank run times it (--phase-events auto): the two events around the matrix pass only; the phases from an
            # instrumented pass behind it (six event records cost a 27-layer step 12 us)
            if args.phase_events == "auto":
                q.pe_mode = "matrix"
            el = None
            for _ in range(2):
                e_ = q.time_steps(max(2, args.warmup), max(5, args.steps))
                if el is None or e_ < el:
                    el = e_
            ph = q.instrumented_phases(max(5, args.steps)) if q.pe_mode == "matrix" else dict(q.phase)
            ent = {"step_ms": el / max(5, args.steps) * 1e3, "cells": q.g.n_cells, "gpus": N,
                   "phase_ms": {k: v / max(5, args.steps) for k, v in ph.items() if v > 0.0}}
            ent["predicted_speedup_step"] = base["step_ms"] / ent["step_ms"]
            qg_ms, _, qg_err = graph_step_ms(q)
            ent["graph_step_ms"] = qg_ms
            if qg_ms and base["graph_step_ms"]:
                ent["predicted_speedup_step_graph"] = base["graph_step_ms"] / qg_ms
            if qg_err:
                ent["graph_error"] = qg_err
            # with the exchange a step adds: one halo sum of b (the step's only exchange)
            best_step = min(x_ for x_ in (ent["step_ms"], qg_ms) if x_)
            best_base = min(x_ for x_ in (base["step_ms"], base["graph_step_ms"]) if x_)
            if xch and ("halo_sum_ms" in xch or "halo_exchange_ms" in xch):   # halo_sum_ms: gather + exchange + scatter-add, as the N-rank step runs them
                ent["predicted_speedup_step_with_exchange"] = best_base / (best_step + xch.get("halo_sum_ms", xch.get("halo_exchange_ms")))
            if dist_cg:
                qc = q.time_cg()
                ent["cg_iteration_ms"] = qc["iteration_ms"]
                ent["predicted_speedup_cg"] = base["cg_iteration_ms"] / ent["cg_iteration_ms"]
                ent["graph_cg_iteration_ms"] = qc["graph_iteration_ms"]
                if qc["graph_iteration_ms"] and base["graph_cg_iteration_ms"]:
                    ent["predicted_speedup_cg_graph"] = base["graph_cg_iteration_ms"] / qc["graph_iteration_ms"]
                # an iteration under a partition: the interface exchange is posted before the whole-domain product and overlaps it (not charged), the two
                # scalar all-reduces sit on the critical path (charged)
                if xch and "allreduce_ms" in xch:
                    bc = min(x_ for x_ in (base["cg_iteration_ms"], base["graph_cg_iteration_ms"]) if x_)
                    qb = min(x_ for x_ in (qc["iteration_ms"], qc["graph_iteration_ms"]) if x_)
                    ent["predicted_speedup_cg_with_exchange"] = bc / (qb + 2 * xch["allreduce_ms"])
            slab_sweep["layers"][str(L_)] = ent
            del q
            torch.cuda.empty_cache()

    if rank == 0:
        K_ = args.steps
        cells_total = n * n * nz_total
        dofs_total = (n + 1) * (n + 1) * (nz_total + 1)          # distinct dofs of the whole mesh (interface dofs, updated on both sides, counted once)
        if args.mesh != "box":
            cells_total, dofs_total = full_grid.n_cells, full_grid.n_nodes
        ms = elapsed / K_ * 1e3
        k_ms = phase["diffusion"] / K_            # fused: the one M + K launch
        # algorithmic bytes of the dominant launch, each datum once: connectivity 32 B + coordinates 24 B per cell, 27 nz × 8 B per
        # cell and matrix written once — 272 B/cell for one matrix, 488 B/cell for the fused pair (2 × 272 = 544 B/cell if each of the two
        # element-integrations were charged its own mesh read: reported as frac_per_integration)
        bytes_per_cell = (BYTES_PER_CELL_MATRIX + 216.0) if fused else BYTES_PER_CELL_MATRIX
        achieved = bytes_per_cell * g.n_cells / (k_ms * 1e-3) / 1e9
        kname = getattr(pr, "matrix_kernel", "") or ("k_matrix_direct<Hex8<2>,DIFFUSION>" if args.strategy != "patch" else "unknown")   # tb_last_kernel_name() after the step's matrix call
        mk_ms = k_ms + phase["mass"] / K_
        # FP64 side of the same launch (SURVEY §8d asks for both fractions).  Flops per cell instance counted on the ISA of the shipped kernel
        # (scripts/isa_hist.py on k_patch_hex8_record<K+M,DIAG>, the same element routine as the staged kernel: 430 v_mul_f64 + 354 v_add_f64 + 465 v_fma_f64 → 1 714 flop); a patch re-integrates
        # the halo cells of its rows, so the kernel executes `instances_per_cell` × that; "useful" charges every cell once.
        inst_per_cell = pr.K.pattern.patch_stats()["instances_per_cell"] if args.strategy == "patch" else 1.0
        import re as _re
        mk_ = _re.match(r"k_patch_hex8_\w+<([^,>]+),([^,>]+)", kname)
        flop_key = (mk_.group(1), mk_.group(2)) if mk_ else None
        flop_inst = FLOP_PER_INSTANCE.get(flop_key, FLOP_PER_INSTANCE[("K+M", "GEN")] if fused else FLOP_PER_INSTANCE[("K", "GEN")])
        tf_useful = flop_inst * g.n_cells / (k_ms * 1e-3) / 1e12
        out = {
            "metric": "element-integrations/sec + DoF-updates/sec, 10M-hex Q1 monodomain",
            "value": 3 * cells_total * K_ / elapsed, "unit": "element-integrations/s",
            "n_gpus": world, "steps": K_, "warmup": args.warmup, "ms_per_step": ms, "host_enqueue_ms_per_step": getattr(pr, "host_enqueue_ms", None), "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "monodomain hot path on %s: assemble M + K (%s scatter, %s) + b (patch-reduced sums, 1.4 global atomics per cell)%s + %s forward-Euler reaction step"
                                   % ({"box": "the %dx%dx%d hex Q1 mesh (%d cells) in %d z-slab(s) of %d layers" % (n, n, nz_total, cells_total, world, part.nzl),
                                       "shuffled": "the %dx%dx%d hex Q1 mesh (%d cells) with cells and nodes RANDOMLY RENUMBERED (unstructured presentation of the same mesh)" % (n, n, nz_total, cells_total),
                                       "lv": "the idealised left-ventricle hex Q1 mesh (%s cells circumferential / radial / longitudinal + O-grid apex: %d cells, %d dofs)" % (args.lv_dims, cells_total, dofs_total)}[args.mesh],
                                      args.strategy, "one fused pass" if fused else "two launches",
                                      (" + neighbour halo sum of b" + ("" if args.no_overlap_exchange else " (b is launched first and its exchange posted behind it; the received partials are added behind the reaction step)")) if world > 1 else "", args.ionic.upper()),
                       "mesh": args.mesh, "renumber": args.renumber,
                       "preroll_ms": args.preroll_ms,   # untimed steps in front of the W warm-up steps: the GPU's clocks (see --preroll-ms)
                       "cells_total": cells_total, "cells_per_gpu": g.n_cells, "dofs_per_gpu": npts, "nnz_per_gpu": sp.nnz, "ionic_model": args.ionic, "ionic_states": ns,
                       "partition": "z-slabs" if gpart is None else "recursive coordinate bisection of the cell centroids (%d parts, this rank: %d neighbour parts, %d shared dofs)"
                                    % (world, len(gpart.neighbours), sum(len(i_) for _, i_ in gpart.neighbours)),
                       "layers_per_gpu": part.nzl if gpart is None else None, **({"backend": "gloo (shared device, test)" if xd is gd else "the C ABI's communicator (tb_comm_*) over the library TB_RCCL_LIBRARY names (shared device, test)"} if share
                          else ({"backend": "nccl (RCCL)" if xd is gd else "RCCL behind the C ABI (tb_comm_*)"} if gd is not None else {}))},
            "dof_updates_per_s": ns * dofs_total * K_ / elapsed,
            # what a host pays ONCE per mesh before the first timed step, outside `value`: host = synthetic mesh + dof table + sparsity pattern + uploads (the
            # generators stand in for Ferrite; a Julia host passes its own arrays), first_step = the plans the first assembly builds (patch decomposition,
            # scatter signatures, one-trip records: host code of tb_plans.cpp) + the first launches
            "setup_s": {"total": t_setup2 - t_setup0, "host_mesh_dofs_pattern_upload": t_setup1 - t_setup0, "first_step_incl_plan_build": t_setup2 - t_setup1,
                        "locality_renumbering": pr.renumber_s, "steps_worth": (t_setup2 - t_setup0) / (elapsed / K_)},
            "pin": FERRITE_PIN,
            **({"checksums": chk} if chk is not None else {}),
            "phase_ms": ({"mass+diffusion": k_ms} if fused else {"mass": phase["mass"] / K_, "diffusion": k_ms}) | {k: phase[k] / K_ for k in ("source", "halo", "reaction")},
            **({"phase_ms_note": phase_note} if phase_note else {}),
            "phase_rates": {"matrix_integrations_per_s": 2 * g.n_cells / (mk_ms * 1e-3),
                            "source_cells_per_s": g.n_cells / (phase["source"] / K_ * 1e-3),
                            "reaction_dof_updates_per_s": ns * npts / (phase["reaction"] / K_ * 1e-3)},
            # "bound" names the roofline `achieved` / `peak` are priced against (algorithmic bytes over the HBM peak, the contract's definition); the FP64
            # vector roof sits beside it in `fp64` (SURVEY §8d asks for both), `nearest_roof` says which of the two fractions is the larger, and
            # `limiter` what the counters and phase stamps say holds the kernel below both (profiles/)
            "roofline": {"kernel": kname, "bound": "hbm",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None, "algorithmic_bytes_per_cell": bytes_per_cell, "launch_ms": k_ms,
                         "frac_per_integration": BYTES_PER_CELL_MATRIX * 2 * g.n_cells / (mk_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "fp64": {"bound": "fp64-valu", "achieved": tf_useful, "achieved_incl_halo": tf_useful * inst_per_cell, "peak": FP64_VECTOR_TFLOPS, "unit": "TFLOP/s",
                                  "frac": tf_useful / FP64_VECTOR_TFLOPS, "frac_incl_halo": tf_useful * inst_per_cell / FP64_VECTOR_TFLOPS,
                                  "flop_per_cell_instance": flop_inst, "flop_count_of_instance": list(flop_key) if flop_key in FLOP_PER_INSTANCE else None, "instances_per_cell": inst_per_cell,
                                  "note": "flops counted on the kernel's ISA (v_mul/v_add = 1, v_fma = 2); useful = one instance per cell, incl_halo = what the kernel executes"},
                         "reaction": {"bound": "hbm", "achieved": (BYTES_PER_DOF_UPDATE + (8 if args.keep_du else 0)) * ns * npts / (phase["reaction"] / K_ * 1e-3) / 1e9,
                                      "peak": HBM_PEAK_GBS, "unit": "GB/s"},
                         "source": {"bound": "hbm", "achieved": BYTES_PER_CELL_VECTOR * g.n_cells / (phase["source"] / K_ * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s"}},
        }
        rf = out["roofline"]
        rf["fp64_frac"], rf["fp64_frac_incl_halo"] = rf["fp64"]["frac"], rf["fp64"]["frac_incl_halo"]
        rf["nearest_roof"] = "fp64-valu" if rf["fp64_frac_incl_halo"] > rf["frac"] else "hbm"
        for k in ("reaction", "source"):
            out["roofline"][k]["frac"] = out["roofline"][k]["achieved"] / HBM_PEAK_GBS
        if ref_rx is not None:
            out["reference_model_reaction"] = ref_rx
        if dist_cg is not None:
            out["distributed_cg"] = dist_cg
            out["cg_iterations_per_s"] = 1e3 / dist_cg["iteration_ms"]
            out["cg_dof_iterations_per_s"] = dofs_total * 1e3 / dist_cg["iteration_ms"]
        if g_ms is not None or g_err is not None:
            out["graph_step"] = {"ms_per_step": g_ms, "nodes": g_nodes, "error": g_err,
                                 "value": None if g_ms is None else 3 * cells_total / (g_ms * 1e-3),
                                 "note": "the same step replayed as ONE HIP-graph launch (tb_graph_begin / _end / _launch; time through the device's time slot), its own "
                                         "timed loop outside the contract's region: `value` above stays the plain-launch figure whose phases the HIP events time"}
        if slab_sweep is not None:
            out["slab_sweep"] = slab_sweep
        try:  # HBM bytes per launch from this round's PMC passes (rocprofv3 cannot run inside this process): scripts/collect_profiles.sh writes the file
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            if tj["cells"] == g.n_cells and args.strategy == "patch":
                kk = tj["kernels"].get("fused" if fused else "diffusion")
                if kk:
                    out["roofline"]["traffic"] = kk["read_bytes"] + kk["write_bytes"]
                    out["roofline"]["traffic_read"] = kk["read_bytes"]
                    out["roofline"]["traffic_write"] = kk["write_bytes"]
                    # what holds the kernel below the HBM roof, from the numbers of this line and the counter file: wasted re-reads if traffic is well above the
                    # algorithmic bytes, otherwise the instruction side (the FP64 fraction the kernel executes, halo included)
                    waste = out["roofline"]["traffic"] / (bytes_per_cell * g.n_cells)
                    out["roofline"]["limiter"] = ("traffic %.2fx the algorithmic bytes, HBM fraction %.2f, FP64 fraction incl. halo %.2f: %s" % (
                        waste, rf["frac"], rf["fp64_frac_incl_halo"],
                        "re-reads / metadata traffic" if waste > 1.5 else "instruction issue and latency (neither roof is near; see the SQ counters of this round's profiles/)"))
                    out["roofline"]["traffic_note"] = ("bytes per launch: reads = 32/64/128-byte TCC_EA0_RDREQ counts x their sizes (FETCH_SIZE tallies 128-B requests "
                                                       "at 64 B on gfx950), writes = WRITE_SIZE; separate rocprofv3 --pmc passes of python3 bench.py, " + tj["source"])
        except Exception:
            pass
        # the plans this numbering got: patch statistics (halo re-integration) and whether the SpMV index compression applies
        if args.strategy == "patch":
            out["patch_stats"] = pr.K.pattern.patch_stats()
        try:
            import ctypes as _C
            o2 = (_C.c_int64 * 2)()
            tb._lib.check(tb.lib().tb_pattern_spmv_plan(pr.K.pattern.h, o2))
            out["spmv_plan"] = {"row_signatures": int(o2[0]), "signature_entries": int(o2[1]), "nnz": int(sp.nnz),
                                "form": "index-compressed (rows share column-offset signatures)" if o2[0] > 0 else "CSR fallback (the numbering shares no signatures)"}
        except Exception as ex:
            out["spmv_plan"] = {"error": str(ex)[:200]}
        if world == 1 and not args.no_cpu_baseline:
            tag = "%dx%dx%d" % (n, n, nz_total) if args.mesh == "box" else "%s(%d cells)" % (args.mesh, g.n_cells)
            big = None if args.no_cpu_big else (g.xyz, g.conn, dh.cell_dofs, dh.ndofs, sp.rowptr, sp.colidx, tag)

            def parity(rp_, ci_, nzM_, nzK_, b_):
                """sampled rows of the device's M, K and b (re-assembled at t = 0) against the oracle's whole-mesh assembly of the cpu_baseline leg"""
                tb.update_operators(pr.M, pr.K, 0.0) if fused else (tb.update_operator(pr.M, 0.0), tb.update_operator(pr.K, 0.0))
                tb.update_operator(pr.src, 0.0)
                torch.cuda.synchronize()
                rows = np.unique(np.concatenate([np.random.default_rng(5).choice(dh.ndofs, 96, replace=False), [0, dh.ndofs - 1]]))
                bh = pr.b.cpu().numpy()
                eM = eK = 0.0
                sM, sK = float(np.abs(nzM_).max()), float(np.abs(nzK_).max())
                for r_ in rows:
                    k0, k1 = int(rp_[r_]), int(rp_[r_ + 1])
                    eM = max(eM, float(np.abs(pr.M.A.view(k0, k1 - k0).to_host() - nzM_[k0:k1]).max()) / sM)
                    eK = max(eK, float(np.abs(pr.K.A.view(k0, k1 - k0).to_host() - nzK_[k0:k1]).max()) / sK)
                eb = float(np.abs(bh - b_).max() / np.abs(b_).max())
                return {"rows_sampled": int(len(rows)), "max_rel_err_M": eM, "max_rel_err_K": eK, "max_rel_err_b_all_entries": eb, "bar": 1e-10,
                        "ok": bool(eM < 1e-10 and eK < 1e-10 and eb < 1e-10)}

            out["cpu_baseline"] = cpu_baseline(args.cpu_n, kap, args.ionic, big=big, check=None if args.no_cpu_big else parity)
        if FERRITE_PIN["note"]:
            sys.stderr.write("bench.py: " + FERRITE_PIN["note"] + "\n")
    # The JSON line is the LAST thing on stdout.  RCCL writes a version banner through C stdio when its first communicator is created; with stdout a pipe
    # that text sits in the C buffer until the process exits — behind the line (seen in round 6: "RCCL version : … Librccl path : …" after the JSON).  So:
    # every rank flushes C stdio, the ranks meet, rank 0 prints, and only then are the communicators taken down.
    if isinstance(xd, tb.distributed.RcclComm):          # the C ABI's communicator goes first: whatever its teardown writes is flushed with the rest
        torch.cuda.synchronize()
        xd.close()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    if dist.is_initialized():
        dist.barrier()
    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
