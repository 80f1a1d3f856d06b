#!/bin/bash
mkdir -p gpurun_out
{
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_ord_known_answer.py -q -x -k "reaction or ord or rush or ORd or float32" 2>&1 | tail -6
for sp in 0 1; do
  echo "== TB_REACTION_SPLIT=$sp"
  TB_REACTION_SPLIT=$sp timeout 600 python3 bench.py --ionic ord --no-cpu-baseline --no-slab-sweep --no-dist-cg --steps 10 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms/step %.3f' % d['ms_per_step'], d['phase_ms'], d['roofline']['reaction'])"
done
} > gpurun_out/gpu_r4_eleventh.log 2>&1
tail -c 4000 gpurun_out/gpu_r4_eleventh.log
