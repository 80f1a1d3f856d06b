#!/bin/bash
# round 6, experiment L: 128-lane workgroups, four per CU (TB_PATCH_T=128: 40 KB of LDS per patch, tiles of 4 x 4 x 4 node layers, 125 instances) against
# the default 256-lane / two-per-CU form
mkdir -p gpurun_out
run() { # label, env... -- bench args
  echo "== $1"; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 900 python3 bench.py --no-slab-sweep --no-graph --no-cpu-baseline --no-dist-cg --steps 20 "$@" 2>gpurun_out/r6l_err.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms/step %.3f' % d['ms_per_step'], {k: round(v,4) for k,v in d['phase_ms'].items()}, d['roofline'].get('kernel'), d.get('patch_stats'))"
  tail -2 gpurun_out/r6l_err.log | cut -c1-300
}
{
timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "synchronising or graph" 2>&1 | tail -3
TB_PATCH_T=128 timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused_mass_diffusion_pair_parity or properties_216 or properties_100_cubed_fused" 2>&1 | tail -5
for rep in 1 2; do
  run t256 TB_X=1 --
  run t128 TB_PATCH_T=128 --
done
run t128-slab TB_PATCH_T=128 -- --layers 27
run t256-slab TB_X=1 -- --layers 27
} > gpurun_out/gpu_r6_l.log 2>&1
cat gpurun_out/gpu_r6_l.log
