#!/bin/bash
# round 6, experiment N: isolated kernel times of the two contraction kernels (one launch each: TB_MECH_CHUNKS=1), kernel trace
mkdir -p gpurun_out
L=$PWD/thunderbolt.jl_amd/libtbhip_ablation.so
export TMPDIR=/tmp
{
for v in c3 c1; do
  echo "== $v"
  export TB_LIBTBHIP=$L TB_MECH_CONTRACT=$v TB_MECH_CHUNKS=1
  rocprofv3 --kernel-trace --stats -d gpurun_out/ktn_$v -o ktn -- python3 scripts/bench_mechanics.py --n 80 --cpu-n 2 --steps 3 > gpurun_out/ktn_$v.json 2>/dev/null
  tail -1 gpurun_out/ktn_$v.json | cut -c1-230
  python3 scripts/rocpd_summary.py gpurun_out/ktn_$v/ktn_results.db 2>/dev/null | head -4 | cut -c1-170
  rm -rf gpurun_out/ktn_$v
done
} > gpurun_out/gpu_r6_n.log 2>&1
cat gpurun_out/gpu_r6_n.log
