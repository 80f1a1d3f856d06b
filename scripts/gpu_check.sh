#!/bin/bash
# One gpurun call that replays what the driver does at round end, in its order: the GPU tests, smoke(), the bench line.
#   /usr/local/graft/bin/gpurun --timeout 3000 -- bash scripts/gpu_check.sh
mkdir -p gpurun_out
{
python3 -m pytest tests -q -m gpu -x 2>&1 | tail -4
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
t0=$SECONDS
python3 bench.py 2> gpurun_out/gpu_check_bench.err | tail -1 > gpurun_out/gpu_check_bench.json
echo "bench.py wall $((SECONDS - t0)) s"
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/gpu_check_bench.json"))
print('ms/step %.3f value %.4g frac %.3f fp64 %.3f/%.3f' % (d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['fp64_frac'], d['roofline']['fp64_frac_incl_halo']), d['phase_ms'])
print('cg', d['cg_iterations_per_s'], d['distributed_cg']['local_spmv_ms'], 'graph_step', (d.get('graph_step') or {}).get('ms_per_step'), 'xch', d['slab_sweep'].get('exchange_latency')); print('kernel', d['roofline']['kernel'], 'limiter', d['roofline'].get('limiter')); print('cpu', d['cpu_baseline']['value'], d['cpu_baseline'].get('parity'))
for k,v in d['slab_sweep']['layers'].items(): print('  ', k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a!='phase_ms'})
print({k:{a:round(b) for a,b in v.items() if a.endswith('per_s')} for k,v in d['cpu_baseline']['table'].items()}, d['cpu_baseline']['physical_cores'])
PY
} > gpurun_out/gpu_check.log 2>&1
tail -c 6000 gpurun_out/gpu_check.log
