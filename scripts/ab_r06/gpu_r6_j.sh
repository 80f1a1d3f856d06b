#!/bin/bash
# round 6, experiment J: the 27-layer slab (the share of one of eight GPUs) under the persistent kernels and the one-patch kernels
mkdir -p gpurun_out
run() { # label, env... -- bench args
  echo "== $1"; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 900 python3 bench.py --no-slab-sweep --no-cpu-baseline --no-dist-cg --steps 50 "$@" 2>gpurun_out/r6j_err.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms/step %.4f' % d['ms_per_step'], {k: round(v,4) for k,v in d['phase_ms'].items()}, d['roofline'].get('kernel'), 'graph', d.get('graph_step',{}).get('ms_per_step'))"
}
{
for rep in 1 2; do
  run "27: record + patch" TB_PATCH_KERNEL=record TB_VECTOR_KERNEL=patch -- --layers 27
  run "27: stream + stream" TB_PATCH_KERNEL=stream -- --layers 27
  run "27: record + stream" TB_PATCH_KERNEL=record -- --layers 27
done
} > gpurun_out/gpu_r6_j.log 2>&1
cat gpurun_out/gpu_r6_j.log
