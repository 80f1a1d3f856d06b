#!/bin/bash
mkdir -p gpurun_out/r5
{
for i in 1 2; do
timeout 600 python3 bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5/bench_async.json
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r5/bench_async.json').read())
print('ms/step %.3f value %.4g'%(d['ms_per_step'], d['value']), d['phase_ms'], 'sum %.3f' % sum(d['phase_ms'].values()), 'graph', d['graph_step']['ms_per_step'])
for L,v in d['slab_sweep']['layers'].items():
    print(L, {k:(round(x,4) if isinstance(x,float) else x) for k,x in v.items() if k.startswith(('step_ms','graph_step','predicted_speedup_step'))})
PY
done
} > gpurun_out/r5/s.log 2>&1
cat gpurun_out/r5/s.log | grep -v amdgpu.ids
