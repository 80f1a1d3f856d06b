#!/bin/bash
# round 4: compressed SpMV + deferred status
mkdir -p gpurun_out
{
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x -k "spmv or deferred or error_codes or heat or cg" 2>&1 | tail -8
timeout 600 python3 -m pytest tests/test_distributed_gpu.py tests/test_rccl_world1.py -q -x 2>&1 | tail -4
TB_PLAN_VERBOSE=1 timeout 900 python3 bench.py --no-cpu-baseline 2> gpurun_out/bench_second.err | tail -1 > gpurun_out/bench_second.json
grep -E "signature|fused patch plan" gpurun_out/bench_second.err | head
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/bench_second.json"))
print("value %.4g ms/step %.3f" % (d["value"], d["ms_per_step"]), d["phase_ms"])
print("cg", d["distributed_cg"]["iteration_ms"], d["distributed_cg"]["local_spmv_ms"], d["distributed_cg"]["product_halo_dot_ms"])
for k,v in d["slab_sweep"]["layers"].items(): print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items()})
print(d["roofline"]["fp64"])
PY
TB_SPMV_KERNEL=rows timeout 600 python3 bench.py --no-cpu-baseline --no-slab-sweep 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rows kernel: cg', d['distributed_cg']['iteration_ms'], d['distributed_cg']['local_spmv_ms'])"
} > gpurun_out/gpu_r4_second.log 2>&1
tail -c 5000 gpurun_out/gpu_r4_second.log
