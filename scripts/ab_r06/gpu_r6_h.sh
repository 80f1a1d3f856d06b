#!/bin/bash
# round 6, experiment H: streaming M + K with batched descriptor reads, non-temporal stores; first-step time with the estimated starting tile;
# config 4 under the scattering strategies (80^3, never reported before)
mkdir -p gpurun_out
run() { # label, env... -- bench args
  echo "== $1"; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 900 python3 bench.py --no-slab-sweep --no-graph --no-cpu-baseline --no-dist-cg --steps 20 "$@" 2>gpurun_out/r6h_err.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms/step %.3f' % d['ms_per_step'], {k: round(v,4) for k,v in d['phase_ms'].items()}, d['roofline'].get('kernel'), 'setup', {k: round(v,2) for k,v in d['setup_s'].items()})"
  grep -E "plan time" gpurun_out/r6h_err.log | head -40
}
{
for rep in 1 2; do
  run record TB_PATCH_KERNEL=record --
  run stream TB_PATCH_KERNEL=stream --
  run stream-nt TB_PATCH_KERNEL=stream TB_LIBTBHIP=$PWD/thunderbolt.jl_amd/libtbhip_nt.so --
done
run plan-times TB_PLAN_VERBOSE=1 --
for st in atomic color; do
  echo "== mechanics 80^3 $st"
  timeout 1200 python3 scripts/bench_mechanics.py --n 80 --strategy $st --steps 3 2>&1 | tail -1 | cut -c1-1500
done
echo "== tets"; timeout 600 python3 scripts/bench_tets.py --n 100 --strategies patch 2>&1 | tail -1 | cut -c1-800
echo "== q2"; timeout 600 python3 scripts/bench_q2_scalar.py --n 64 --strategies element 2>&1 | tail -1 | cut -c1-800
} > gpurun_out/gpu_r6_h.log 2>&1
cat gpurun_out/gpu_r6_h.log
