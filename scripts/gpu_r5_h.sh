#!/bin/bash
mkdir -p gpurun_out/r5
R=$GRAFT_REPO_ROOT
{
for mi in 320 400 512; do
  echo "== TB_RCB_MAX_INST=$mi"
  TB_LIBTBHIP=$R/thunderbolt.jl_amd/libtbhip_ablation.so TB_RCB_MAX_INST=$mi TB_PLAN_VERBOSE=1 timeout 600 python3 bench.py --mesh lv --steps 20 --no-cpu-baseline --no-dist-cg 2> /tmp/lv.err | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('  ms/step %.3f' % d['ms_per_step'], d['phase_ms'], d['patch_stats']['patches'], d['patch_stats']['max_instances'])"
  grep "patch plan:" /tmp/lv.err | tail -2
done
} > gpurun_out/r5/h.log 2>&1
cat gpurun_out/r5/h.log
