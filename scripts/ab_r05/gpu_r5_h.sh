#!/bin/bash
mkdir -p gpurun_out/r5
{
TB_PLAN_VERBOSE=1 timeout 600 python3 bench.py --mesh lv --steps 20 --no-cpu-baseline --no-dist-cg 2> /tmp/lv.err | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('  ms/step %.3f' % d['ms_per_step'], d['phase_ms'], d['patch_stats'])"
grep "patch plan" /tmp/lv.err | tail -6
python3 -m pytest tests/test_gpu_parity.py -q -x -k "bisection or config5 or unstructured or source" 2>&1 | tail -3
} > gpurun_out/r5/h.log 2>&1
cat gpurun_out/r5/h.log
