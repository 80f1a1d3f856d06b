#!/bin/bash
mkdir -p gpurun_out/r5
{
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x -k "reaction or rush or single_cell or graph_replay" 2>&1 | tail -5
for i in 1 2; do
timeout 300 python3 bench.py --steps 20 --no-cpu-baseline --no-dist-cg --no-slab-sweep --no-graph 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('stream ms/step %.3f' % d['ms_per_step'], d['phase_ms'])"
TB_LIBTBHIP=$PWD/thunderbolt.jl_amd/libtbhip_ablation.so TB_REACTION_KERNEL=plain timeout 300 python3 bench.py --steps 20 --no-cpu-baseline --no-dist-cg --no-slab-sweep --no-graph 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('plain  ms/step %.3f' % d['ms_per_step'], d['phase_ms'])"
done
} > gpurun_out/r5/l.log 2>&1
cat gpurun_out/r5/l.log | grep -v amdgpu.ids
