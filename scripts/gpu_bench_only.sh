#!/bin/bash
# the bench line alone, with its wall time (what the driver's clock sees around `python bench.py`)
mkdir -p gpurun_out
{
t0=$SECONDS
python3 bench.py 2> gpurun_out/gpu_check_bench.err | tail -1 > gpurun_out/gpu_check_bench.json
echo "bench.py wall $((SECONDS - t0)) s"
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/gpu_check_bench.json"))
print('ms/step %.3f value %.4g frac %.3f fp64 %.3f/%.3f' % (d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['fp64_frac'], d['roofline']['fp64_frac_incl_halo']), d['phase_ms'])
print('cg', d['cg_iterations_per_s'], d['distributed_cg']['local_spmv_ms'])
for k,v in d['slab_sweep']['layers'].items(): print('  ', k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a!='phase_ms'})
print({k:{a:round(b) for a,b in v.items() if a.endswith('per_s')} for k,v in d['cpu_baseline']['table'].items()}, d['cpu_baseline']['physical_cores'])
PY
} > gpurun_out/gpu_bench_only.log 2>&1
tail -c 5000 gpurun_out/gpu_bench_only.log
