#!/bin/bash
# round 4: balanced tile cuts, balanced reaction grid; stamps of the patch kernels
mkdir -p gpurun_out
{
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -q -x -k "matrix_parity or fused or pair or 216 or source or tet or reaction or strateg or loaded or element" 2>&1 | tail -6
for cut in full balanced; do
  echo "== TB_PATCH_CUT=$cut"
  TB_PLAN_VERBOSE=1 TB_PATCH_CUT=$cut timeout 900 python3 bench.py --no-cpu-baseline --steps 20 2> gpurun_out/b5.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms/step %.3f' % d['ms_per_step'], d['phase_ms'], 'cg', round(d['distributed_cg']['iteration_ms'],4))
for k,v in d['slab_sweep']['layers'].items(): print('  ', k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items()})"
  grep -E "fused patch plan|vector patch plan" gpurun_out/b5.err | head -3
done
bash scripts/gpu_r4_stamps.sh
} > gpurun_out/gpu_r4_fifth.log 2>&1
tail -c 7000 gpurun_out/gpu_r4_fifth.log
