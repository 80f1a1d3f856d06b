#!/bin/bash
# round 4 checkpoint: whole GPU tier, then the bench line
mkdir -p gpurun_out
{
timeout 3000 python3 -m pytest tests -q -m gpu -x 2>&1 | tail -8
TB_PLAN_VERBOSE=1 timeout 900 python3 bench.py --steps 20 2> gpurun_out/b6.err | tail -1 > gpurun_out/bench_sixth.json
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/bench_sixth.json"))
print('ms/step %.3f value %.4g' % (d['ms_per_step'], d['value']), d['phase_ms'], 'cg', d['distributed_cg'])
for k,v in d['slab_sweep']['layers'].items(): print('  ', k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items()})
print(d['roofline']['frac'], d['roofline']['fp64'])
print({k:{a:round(b) for a,b in v.items() if a.endswith('per_s')} for k,v in d['cpu_baseline']['table'].items()}, d['cpu_baseline']['physical_cores'])
PY
grep -E "fused patch plan|vector patch plan|signature plan|records" gpurun_out/b6.err | head -4
} > gpurun_out/gpu_r4_sixth.log 2>&1
tail -c 6000 gpurun_out/gpu_r4_sixth.log
