#!/bin/bash
# round 6, experiment O: the linear form inside the fused M + K pass (tb_assemble_matrix_pair_vector) against the separate vector launch
mkdir -p gpurun_out
run() { # label, env... -- bench args
  echo "== $1"; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 900 python3 bench.py --no-slab-sweep --no-cpu-baseline --no-dist-cg --steps 20 "$@" 2>gpurun_out/r6o_err.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms/step %.4f' % d['ms_per_step'], {k: round(v,4) for k,v in d['phase_ms'].items()}, d['roofline'].get('kernel'), 'graph', d.get('graph_step',{}).get('ms_per_step'), 'frac', round(d['roofline']['frac'],4))"
  tail -1 gpurun_out/r6o_err.log | cut -c1-200
}
{
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "triple or fused_mass_diffusion_pair or source_vector or graph_replay or patch_kernel_variants" 2>&1 | tail -5
for rep in 1 2; do
  run separate TB_X=1 -- --no-fuse-source
  run fused-b TB_X=1 --
done
run separate-27 TB_X=1 -- --no-fuse-source --layers 27 --steps 50
run fused-b-27 TB_X=1 -- --layers 27 --steps 50
run fused-b-lv TB_X=1 -- --mesh lv
run separate-lv TB_X=1 -- --mesh lv --no-fuse-source
} > gpurun_out/gpu_r6_o.log 2>&1
cat gpurun_out/gpu_r6_o.log
