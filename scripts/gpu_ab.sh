#!/bin/bash
# A/B runs of the bench kernels through the library's environment switches, one gpurun call:
#   gpurun -- bash scripts/gpu_ab.sh "TB_PATCH_KERNEL=staged" "TB_PATCH_KERNEL=record" "TB_SPMV_KERNEL=wave" …
# every argument is one environment assignment list (quoted); each is run twice, interleaved, so that box drift shows.
mkdir -p gpurun_out
{
for rep in 1 2; do
  for envs in "$@"; do
    echo "== $envs"
    env $envs timeout 600 python3 bench.py --no-cpu-baseline --no-slab-sweep --steps 20 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms/step %.3f' % d['ms_per_step'], d['phase_ms'], 'cg', round(d['distributed_cg']['iteration_ms'],4), round(d['distributed_cg']['local_spmv_ms'],4))"
  done
done
} > gpurun_out/gpu_ab.log 2>&1
tail -c 4000 gpurun_out/gpu_ab.log
