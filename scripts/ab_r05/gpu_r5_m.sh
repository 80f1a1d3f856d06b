#!/bin/bash
mkdir -p gpurun_out/r5
{
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x -k "source or linear_form or config1 or bisection or vector or graph_replay or unstructured" 2>&1 | tail -5
for i in 1 2; do
timeout 300 python3 bench.py --steps 20 --no-cpu-baseline --no-dist-cg --no-slab-sweep --no-graph 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('record ms/step %.3f' % d['ms_per_step'], d['phase_ms'])"
TB_LIBTBHIP=$PWD/thunderbolt.jl_amd/libtbhip_ablation.so timeout 300 python3 bench.py --steps 20 --no-cpu-baseline --no-dist-cg --no-slab-sweep --no-graph 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('old    ms/step %.3f' % d['ms_per_step'], d['phase_ms'])"
done
python3 scripts/slab_kernels.py 27
timeout 300 python3 bench.py --mesh lv --steps 20 --no-cpu-baseline --no-dist-cg --no-graph 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('lv ms/step %.3f' % d['ms_per_step'], d['phase_ms'])"
} > gpurun_out/r5/m.log 2>&1
cat gpurun_out/r5/m.log | grep -v amdgpu.ids
