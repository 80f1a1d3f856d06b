"""The `nccl` (= RCCL) code path, executed on a one-GPU box (missing 2): a child process under `torch.distributed.run --nproc-per-node 1`
initialises RCCL at world size 1 and runs the halo exchange, the all-reduces and the overlapped CG iteration through their device-buffer branches
(tests/rccl_world1_child.py); bench.py under the same launcher creates its `nccl` process group, barriers and max-reduces its timing at world size 1.
The parent only starts children (no exec of this GPU-initialised process)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _torchrun(script_args, timeout=900, env=None):
    env = dict(os.environ if env is None else env)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "--"] + script_args
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)


def test_rccl_world_size_one_device_buffer_branches():
    r = _torchrun([os.path.join(ROOT, "tests", "rccl_world1_child.py")])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and lines, (r.stdout[-3000:], r.stderr[-3000:])
    d = json.loads(lines[-1])
    assert d["ok"] and d["backend"] == "nccl" and d["world"] == 1, d
    assert d["halo_device_exchange_exact"] and d["torch_indexing_exchange_equal"]
    assert d["cg_product_err"] < 1e-13 and d["cg_steps_finite_and_flag_clear"]


def test_rccl_behind_the_c_abi_world_size_one():
    """tb_comm_* (RCCL opened by libtbhip itself, no torch.distributed anywhere): communicator at world size 1, the grouped send / receive of the halo
    exchange with rank 0 as its own neighbour, sum and max all-reduces, the overlapped CG iteration — what a Julia host without GPU-aware MPI would call."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_world1_child.py"), "abi"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and lines, (r.stdout[-3000:], r.stderr[-3000:])
    d = json.loads(lines[-1])
    assert d["ok"] and d["backend"] == "tbhip-rccl" and d["world"] == 1, d
    assert d["halo_device_exchange_exact"] and d["halo_second_exchange_exact"] and d["all_reduce_device"] and d["all_reduce_max"]
    assert d["cg_product_err"] < 1e-13 and d["cg_steps_finite_and_flag_clear"]


@pytest.mark.parametrize("exchange,backend", [("torch", "nccl (RCCL)"), ("abi", "RCCL behind the C ABI (tb_comm_*)"), ("auto", "RCCL behind the C ABI (tb_comm_*)")])
def test_bench_under_launcher_initialises_rccl_at_world_size_one(exchange, backend):
    """bench.py under the launcher at world size 1: process group, barrier, max-reduced timings; with --exchange abi the communicator of the C ABI is created
    from the id that torch.distributed broadcasts, and the CG reductions / halo exchange go through tb_comm_*."""
    r = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--n", "24", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-slab-sweep", "--exchange", exchange])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-3000:], r.stderr[-3000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["backend"] == backend and d["value"] > 0 and d["cg_iterations_per_s"] > 0
    # the line is the LAST thing on stdout: RCCL's version banner (C stdio, buffered on a pipe until exit) is flushed in front of it
    assert [ln for ln in r.stdout.splitlines() if ln.strip()][-1].startswith("{"), r.stdout[-1500:]


def test_bench_exchange_auto_falls_back_to_torch_distributed_when_no_rank_can_create_the_communicator():
    """--exchange auto: a rank that cannot create the C ABI's communicator votes 0, the minimum over the ranks decides, torch.distributed carries the data"""
    env = dict(os.environ, TB_BENCH_NO_ABI_COMM="1")
    r = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--n", "24", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-slab-sweep"], env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-3000:], r.stderr[-3000:])
    d = json.loads(lines[0])
    assert d["config"]["backend"] == "nccl (RCCL)" and "torch.distributed carries the exchange" in r.stderr
    r = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--n", "24", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-slab-sweep", "--exchange", "abi"], env=env)
    assert r.returncode != 0 and "could not be created on every rank" in r.stderr


@pytest.mark.parametrize("mesh_args,own_queue", [(["--n", "24"], False), (["--mesh", "lv", "--lv-dims", "24,4,16"], False), (["--n", "24"], True)])
def test_three_ranks_through_the_c_abi_communicator_over_the_test_double(tmp_path, mesh_args, own_queue):
    """The C ABI's multi-rank path on a one-GPU box: RCCL refuses two ranks on one device, so tests/mock_rccl (a test double for the nine entry points
    tb_comm.cpp binds: messages staged through shared memory) stands in for it — TB_RCCL_LIBRARY.  Three ranks of bench.py on cuda:0: the communicator id
    travels over torch.distributed (gloo), tb_comm_create joins three ranks, every halo sum is one tb_comm_exchange with the rank's real neighbour list (the
    middle rank: two peers in one group; the ventricle under the bisection partition: whatever parts touch), the CG reductions are tb_comm_allreduce —
    and the run leaves the one-rank sums (1e-10).  What this checks is OUR
    side of every call (peers, counts, pointers, grouping, order); RCCL's own behaviour between GPUs stays untested here."""
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    so = str(tmp_path / "libmockrccl.so")
    b = subprocess.run([hipcc, "-O1", "-shared", "-fPIC", "-I/opt/rocm/include", "-o", so, os.path.join(ROOT, "tests", "mock_rccl", "mock_rccl.cpp"), "-lrt"],
                       capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-2000:]
    common = mesh_args + ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-graph", "--no-slab-sweep", "--checksums", "--preroll-ms", "0"]

    def line(extra, env):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra + common, env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])

    one = line(["--gpus", "1"], {})
    # own_queue: the step's exchange through tb_comm_exchange_begin / _end (the communicator's queue, joined by events) instead of the device's queue
    three = line(["--gpus", "3", "--exchange", "abi"], dict({"TB_BENCH_SHARE_DEVICE": "1", "TB_RCCL_LIBRARY": so}, **({"TB_BENCH_HALO_OWN_QUEUE": "1"} if own_queue else {})))
    assert three["n_gpus"] == 3 and "tb_comm_*" in three["config"]["backend"] and three["phase_ms"]["halo"] > 0.0
    assert three["cg_iterations_per_s"] > 0 and three["distributed_cg"]["halo_bytes_per_rank"] > 0
    for k, v in one["checksums"].items():
        assert abs(three["checksums"][k] - v) <= 1e-10 * abs(v), (k, v, three["checksums"][k])
