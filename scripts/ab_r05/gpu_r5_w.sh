#!/bin/bash
mkdir -p gpurun_out/r5
{
for pe in all matrix none; do
timeout 600 python3 bench.py --no-cpu-baseline --no-dist-cg --phase-events $pe 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$pe ms/step %.3f' % d['ms_per_step'], 'graph', round(d['graph_step']['ms_per_step'],3), {L:(round(v['step_ms'],4), round(v.get('graph_step_ms',0),4)) for L,v in d['slab_sweep']['layers'].items()})"
done
} > gpurun_out/r5/w.log 2>&1
grep -v amdgpu.ids gpurun_out/r5/w.log
