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
