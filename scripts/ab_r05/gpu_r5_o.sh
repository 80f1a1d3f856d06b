#!/bin/bash
mkdir -p gpurun_out/r5
export TB_LIBTBHIP=$PWD/thunderbolt.jl_amd/libtbhip_ablation.so
{
for mi in 256 320 384 448 512 640; do
echo "== TB_RCB_MAX_INST=$mi"
TB_RCB_MAX_INST=$mi TB_PLAN_VERBOSE=1 timeout 600 python3 bench.py --mesh lv --steps 20 --no-cpu-baseline --no-dist-cg --no-graph 2> /tmp/lv.err | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('  ms/step %.3f' % d['ms_per_step'], d['phase_ms']['mass+diffusion'], d['patch_stats'])"
grep "patch plan: tiles" /tmp/lv.err | tail -1
done
} > gpurun_out/r5/o.log 2>&1
cat gpurun_out/r5/o.log | grep -v amdgpu.ids
