#!/bin/bash
# round 6, experiment G: tile-major dof numbering (TB_ORDER_TILES: a patch's rows are one run of consecutive dofs) on the box; plan-stage times
mkdir -p gpurun_out
run() { # label, env... -- bench args
  echo "== $1"; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 900 python3 bench.py --no-slab-sweep --no-graph --no-cpu-baseline --steps 20 "$@" 2>gpurun_out/r6g_err.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms/step %.3f' % d['ms_per_step'], {k: round(v,4) for k,v in d['phase_ms'].items()}, d['roofline'].get('kernel'), 'cg', round(d['distributed_cg']['iteration_ms'],4), round(d['distributed_cg']['local_spmv_ms'],4), 'setup', {k: round(v,2) for k,v in d['setup_s'].items()}, d.get('spmv_plan'), d.get('patch_stats'))"
  grep -E "plan time" gpurun_out/r6g_err.log | head -60
}
{
run box-record TB_PATCH_KERNEL=record TB_PLAN_VERBOSE=1 --
run box-tiles-dofs-record TB_PATCH_KERNEL=record -- --renumber dofs --renumber-order tiles
run box-tiles-grid-record TB_PATCH_KERNEL=record -- --renumber grid --renumber-order tiles
run box-tiles-grid-stream TB_PATCH_KERNEL=stream -- --renumber grid --renumber-order tiles
run box-stream TB_PATCH_KERNEL=stream --
} > gpurun_out/gpu_r6_g.log 2>&1
cat gpurun_out/gpu_r6_g.log
