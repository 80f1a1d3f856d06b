#!/bin/bash
mkdir -p gpurun_out
{
python3 -m pytest tests/test_gpu_parity.py tests/test_distributed_gpu.py tests/test_rccl_world1.py -q -x -k "sliced_mirror or index_compressed or monodomain or config2 or config3 or heat or euler or distributed or rccl or bench" 2>&1 | tail -5
python3 examples/monodomain_fhn.py --ionic tt06 --n 216 2>/dev/null | tail -1
python3 bench.py --no-slab-sweep --no-cpu-big 2> gpurun_out/mirror_bench.err | tail -1 > gpurun_out/mirror_bench.json
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/mirror_bench.json"))
print('ms/step %.3f cg/s %.1f' % (d['ms_per_step'], d['cg_iterations_per_s']))
print({k:v for k,v in d['distributed_cg'].items() if k!='note'})
PY
} > gpurun_out/r4_mirror.log 2>&1
tail -c 5000 gpurun_out/r4_mirror.log
