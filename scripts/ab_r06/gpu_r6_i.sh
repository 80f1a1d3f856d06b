#!/bin/bash
# round 6, experiment I: quarter-wave write-out in the record kernel (TB_PATCH_WQ=0: the half-wave form), against the streaming kernel; tets with the
# quarter-wave write-out
mkdir -p gpurun_out
L=$PWD/thunderbolt.jl_amd/libtbhip_ablation.so
run() { # label, env... -- bench args
  echo "== $1"; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env TB_LIBTBHIP=$L "${envs[@]}" timeout 900 python3 bench.py --no-slab-sweep --no-graph --no-cpu-baseline --no-dist-cg --steps 20 "$@" 2>gpurun_out/r6i_err.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms/step %.3f' % d['ms_per_step'], {k: round(v,4) for k,v in d['phase_ms'].items()}, d['roofline'].get('kernel'))"
  grep -E "phases" gpurun_out/r6i_err.log | tail -1
}
{
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused or patch_kernel_variants or unstructured or tet" 2>&1 | tail -3
for rep in 1 2; do
  run record-wq1 TB_PATCH_KERNEL=record --
  run record-wq0 TB_PATCH_KERNEL=record TB_PATCH_WQ=0 --
  run stream TB_PATCH_KERNEL=stream --
done
run stamps-record-wq1 TB_PATCH_KERNEL=record TB_PROF_STAMPS=1 --
run stamps-record-wq0 TB_PATCH_KERNEL=record TB_PATCH_WQ=0 TB_PROF_STAMPS=1 --
echo "== tets"; timeout 600 python3 scripts/bench_tets.py --n 100 --strategies patch 2>&1 | tail -1 | cut -c1-800
} > gpurun_out/gpu_r6_i.log 2>&1
cat gpurun_out/gpu_r6_i.log
