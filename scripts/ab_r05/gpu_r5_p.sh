#!/bin/bash
mkdir -p gpurun_out/r5
{
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_distributed_gpu.py tests/test_rccl_world1.py -q -x -k "cg or CG or heat or solve or spmv or distributed or rccl or newton or gmres or mechanics_solve or graph" 2>&1 | tail -5
timeout 600 python3 bench.py --steps 20 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5/bench_slots.json
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r5/bench_slots.json').read())
print('ms/step %.3f'%d['ms_per_step'], d['phase_ms'])
print('cg', {k:v for k,v in d['distributed_cg'].items() if 'ms' in k or 'iteration' in k})
for L,v in d['slab_sweep']['layers'].items():
    print(L, {k:(round(x,4) if isinstance(x,float) else x) for k,x in v.items() if k.startswith(('cg','graph_cg','predicted_speedup_cg','step_ms','graph_step','predicted_speedup_step'))})
PY
} > gpurun_out/r5/p.log 2>&1
cat gpurun_out/r5/p.log | grep -v amdgpu.ids
