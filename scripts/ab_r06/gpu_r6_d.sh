#!/bin/bash
# round 6, experiment D: streaming kernel with ticket-dealt patches; locality renumbering of the shuffled box; host probe
mkdir -p gpurun_out
L=$PWD/thunderbolt.jl_amd/libtbhip_ablation.so
run() { # label, env... -- bench args
  echo "== $1"; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 900 python3 bench.py --no-cpu-baseline --no-slab-sweep --no-graph --steps 20 "$@" 2>gpurun_out/r6d_err.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms/step %.3f' % d['ms_per_step'], {k: round(v,4) for k,v in d['phase_ms'].items()}, d['roofline'].get('kernel'), 'cg', round(d['distributed_cg']['iteration_ms'],4) if 'distributed_cg' in d else None, 'setup', {k: round(v,2) for k,v in d['setup_s'].items()}, d.get('spmv_plan',{}).get('form'), d.get('patch_stats'))"
  grep -E "phases" gpurun_out/r6d_err.log | tail -1
}
{
bash scripts/host_probe.sh
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused or patch_kernel_variants" 2>&1 | tail -3
for rep in 1 2; do
  run record TB_PATCH_KERNEL=record -- --no-dist-cg
  run stream TB_PATCH_KERNEL=stream -- --no-dist-cg
done
run stamps-stream TB_LIBTBHIP=$L TB_PROF_STAMPS=1 TB_PATCH_KERNEL=stream -- --no-dist-cg
run box TB_X=1 --
run shuffled TB_X=1 -- --mesh shuffled
run shuffled-dofs TB_X=1 -- --mesh shuffled --renumber dofs
run shuffled-grid TB_X=1 -- --mesh shuffled --renumber grid
} > gpurun_out/gpu_r6_d.log 2>&1
cat gpurun_out/gpu_r6_d.log
