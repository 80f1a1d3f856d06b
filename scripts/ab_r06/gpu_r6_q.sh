#!/bin/bash
# round 6, experiment Q: does the OpenMP binding of the main thread (libgomp pins it to the first core) cost the one-rank timed loop anything?  And the 27-layer slab,
# where the host's enqueue time matters most; two ranks on one device with and without the inherited binding.
mkdir -p gpurun_out
run() { # label, env... -- bench args
  echo "== $1"; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 900 python3 bench.py --no-slab-sweep --no-cpu-baseline --no-dist-cg --steps 50 "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms/step %.4f' % d['ms_per_step'], {k: round(v,4) for k,v in d['phase_ms'].items()}, 'graph', d.get('graph_step',{}).get('ms_per_step'))"
}
{
for rep in 1 2; do
  run "pinned (default at N = 1)" TB_X=1 --
  run "unpinned" OMP_PROC_BIND=false --
  run "pinned, 27 layers" TB_X=1 -- --layers 27
  run "unpinned, 27 layers" OMP_PROC_BIND=false -- --layers 27
done
run "2 ranks one device, 24^3" TB_BENCH_SHARE_DEVICE=1 -- --gpus 2 --n 48
run "2 ranks one device, 24^3, binding forced on the ranks (the old behaviour)" TB_BENCH_SHARE_DEVICE=1 OMP_PROC_BIND=close OMP_PLACES=cores -- --gpus 2 --n 48
} > gpurun_out/gpu_r6_q.log 2>&1
cat gpurun_out/gpu_r6_q.log
