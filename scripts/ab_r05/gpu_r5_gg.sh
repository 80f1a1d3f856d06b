#!/bin/bash
mkdir -p gpurun_out/r5
{
for pr in 150 1000 4000 150 4000; do
timeout 600 python3 bench.py --no-cpu-baseline --no-dist-cg --no-slab-sweep --no-graph --preroll-ms $pr 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('preroll $pr ms: ms/step %.3f' % d['ms_per_step'], {k:round(v,4) for k,v in d['phase_ms'].items()})"
done
} > gpurun_out/r5/gg.log 2>&1
grep -v amdgpu.ids gpurun_out/r5/gg.log
