#!/bin/bash
# round 6, experiment A: does the fused M + K kernel lose time to two co-resident workgroups integrating at half rate each?
#   pad   : one workgroup per CU (LDS request > 80 KB) with phase stamps: the integration phase of a workgroup that has its SIMDs to itself
#   prio  : asymmetric integration priority by LDS slot (TB_PATCH_PRIO=2/3/4)
mkdir -p gpurun_out
L=$PWD/thunderbolt.jl_amd/libtbhip_ablation.so
run() { # label, env...
  echo "== $1"; shift
  env TB_LIBTBHIP=$L "$@" timeout 600 python3 bench.py --no-cpu-baseline --no-slab-sweep --no-dist-cg --no-graph --steps 20 2>gpurun_out/r6a_err.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms/step %.3f' % d['ms_per_step'], d['phase_ms'], d['roofline']['launch_ms'] if 'launch_ms' in d['roofline'] else '')"
  grep -E "staged phases" gpurun_out/r6a_err.log | tail -1
}
{
for rep in 1 2; do
  run base TB_X=1
  run prio2 TB_PATCH_PRIO=2
  run prio3 TB_PATCH_PRIO=3
  run prio4 TB_PATCH_PRIO=4
done
run stamps-2wg TB_PROF_STAMPS=1
run stamps-1wg TB_PROF_STAMPS=1 TB_PATCH_LDS_PAD=100000
run stamps-prio3 TB_PROF_STAMPS=1 TB_PATCH_PRIO=3
} > gpurun_out/gpu_r6_a.log 2>&1
cat gpurun_out/gpu_r6_a.log
