#!/bin/bash
mkdir -p gpurun_out/r5
{
for pe in all matrix none all matrix none; do
timeout 600 python3 bench.py --no-cpu-baseline --no-dist-cg --no-slab-sweep --phase-events $pe 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$pe ms/step %.3f' % d['ms_per_step'], d['phase_ms'], 'graph', d['graph_step']['ms_per_step'])"
done
export TMPDIR=/tmp; cd /tmp; rocprofv3 --kernel-trace --stats -d /tmp/ktx -o ktx -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-dist-cg --no-slab-sweep --no-graph --steps 20 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python3 scripts/rocpd_summary.py /tmp/ktx/ktx_results.db | head -14 | cut -c1-160
} > gpurun_out/r5/t.log 2>&1
cat gpurun_out/r5/t.log | grep -v amdgpu.ids
