#!/bin/bash
# round 6, experiment F: what bounds the fused kernel — work-skipping switches of the profiling build on both forms (1: no LDS adds, 4: no arithmetic, 5: neither)
mkdir -p gpurun_out
L=$PWD/thunderbolt.jl_amd/libtbhip_ablation.so
run() { # label, env... -- bench args
  echo "== $1"; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env TB_LIBTBHIP=$L "${envs[@]}" timeout 900 python3 bench.py --no-slab-sweep --no-graph --no-dist-cg --no-cpu-baseline --steps 20 "$@" 2>gpurun_out/r6f_err.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms/step %.3f' % d['ms_per_step'], {k: round(v,4) for k,v in d['phase_ms'].items()}, d['roofline'].get('kernel'))"
  grep -E "phases" gpurun_out/r6f_err.log | tail -1
}
{
for k in record stream; do
  for f in 0 1 4 5; do
    run "$k flags=$f" TB_PATCH_KERNEL=$k TB_DEBUG_FLAGS=$f TB_PROF_STAMPS=1 --
  done
done
} > gpurun_out/gpu_r6_f.log 2>&1
cat gpurun_out/gpu_r6_f.log
