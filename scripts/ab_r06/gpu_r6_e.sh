#!/bin/bash
# round 6, experiment E: streaming kernels (M + K: vector header load, async ticket, quarter-wave write-out; source vector: streaming form) A/B
mkdir -p gpurun_out
L=$PWD/thunderbolt.jl_amd/libtbhip_ablation.so
run() { # label, env... -- bench args
  echo "== $1"; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 900 python3 bench.py --no-slab-sweep --no-graph --steps 20 "$@" 2>gpurun_out/r6e_err.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms/step %.3f' % d['ms_per_step'], {k: round(v,4) for k,v in d['phase_ms'].items()}, d['roofline'].get('kernel'), 'setup', {k: round(v,2) for k,v in d['setup_s'].items()}); cb=d.get('cpu_baseline'); print('  cpu', cb['value'], cb['cores'], cb.get('thread_efficiency'), {k:{a:(round(b) if isinstance(b,float) else b) for a,b in v.items() if a in ('element_integrations_per_s','element_integrations_per_s_median')} for k,v in cb['table'].items()}) if cb else None"
  grep -E "phases" gpurun_out/r6e_err.log | tail -1
}
{
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused or patch_kernel_variants or source or linear_form or config1 or vector" 2>&1 | tail -3
for rep in 1 2; do
  run record+patch TB_PATCH_KERNEL=record TB_VECTOR_KERNEL=patch -- --no-dist-cg --no-cpu-baseline
  run stream+stream TB_PATCH_KERNEL=stream -- --no-dist-cg --no-cpu-baseline
done
run stamps-stream TB_LIBTBHIP=$L TB_PROF_STAMPS=1 TB_PATCH_KERNEL=stream -- --no-dist-cg --no-cpu-baseline
run full-line TB_X=1 --
} > gpurun_out/gpu_r6_e.log 2>&1
cat gpurun_out/gpu_r6_e.log
