#!/bin/bash
mkdir -p gpurun_out
{
timeout 1800 python3 -m pytest tests/test_gpu_parity.py -q -x -k "config5 or variants" 2>&1 | tail -6
for pr in 0 1; do
  echo "== TB_PATCH_PRIO=$pr"
  TB_PATCH_PRIO=$pr timeout 600 python3 bench.py --no-cpu-baseline --no-slab-sweep --steps 20 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms/step %.3f' % d['ms_per_step'], d['phase_ms'])"
done
echo "== electromechanics"
timeout 900 python3 scripts/bench_electromechanics.py 2>&1 | tail -2
timeout 900 python3 scripts/bench_electromechanics.py --order 2 --nc 64 --nr 4 --nl 50 2>&1 | tail -2
} > gpurun_out/gpu_r4_seventh.log 2>&1
tail -c 6000 gpurun_out/gpu_r4_seventh.log
