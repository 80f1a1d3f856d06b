#!/bin/bash
mkdir -p gpurun_out/r5
{
for pr in 150 0 150 0; do
t0=$SECONDS
timeout 600 python3 bench.py --no-cpu-baseline --preroll-ms $pr 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('preroll $pr ms/step %.3f value %.4g frac %.3f' % (d['ms_per_step'], d['value'], d['roofline']['frac']), {k:round(v,4) for k,v in d['phase_ms'].items()}, 'graph', round(d['graph_step']['ms_per_step'],3), 'cg', round(d['distributed_cg']['iteration_ms'],4), {L:(round(v['step_ms'],4), round(v.get('graph_step_ms',0),4), round(v['cg_iteration_ms'],4)) for L,v in d['slab_sweep']['layers'].items()})"
echo "  wall $((SECONDS-t0)) s"
done
} > gpurun_out/r5/y.log 2>&1
grep -v amdgpu.ids gpurun_out/r5/y.log
