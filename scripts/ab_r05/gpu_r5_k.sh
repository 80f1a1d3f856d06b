#!/bin/bash
mkdir -p gpurun_out/r5
{
python3 -m pytest tests/test_gpu_parity.py -q -x -k "fused or patch or heat or config2 or config3 or 216" 2>&1 | tail -3
for i in 1 2; do
python3 bench.py --steps 20 --no-cpu-baseline --no-dist-cg --no-slab-sweep 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('product  ms/step %.3f' % d['ms_per_step'], d['phase_ms']['mass+diffusion'], d['roofline']['kernel'])"
TB_LIBTBHIP=$PWD/thunderbolt.jl_amd/libtbhip_ablation.so TB_PATCH_POS8=0 python3 bench.py --steps 20 --no-cpu-baseline --no-dist-cg --no-slab-sweep 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('pos8 off ms/step %.3f' % d['ms_per_step'], d['phase_ms']['mass+diffusion'], d['roofline']['kernel'])"
done
} > gpurun_out/r5/k.log 2>&1
cat gpurun_out/r5/k.log | grep -v amdgpu.ids
