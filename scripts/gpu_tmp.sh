#!/bin/bash
mkdir -p gpurun_out
export TB_LIBTBHIP=$PWD/thunderbolt.jl_amd/libtbhip_ablation.so
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
{
for fl in 0 1 2; do
  echo "== TB_DEBUG_FLAGS=$fl"
  export TB_DEBUG_FLAGS=$fl
  rm -rf /tmp/ab$fl
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ab$fl -o m -- python3 $R/scripts/bench_mechanics.py --n 80 --steps 2 --cpu-n 2 > /tmp/ab$fl.log 2>&1
  python3 $R/scripts/rocpd_summary.py $(find /tmp/ab$fl -name "*.db" | head -1) | head -3 | cut -c1-160
done
} > $R/gpurun_out/tmp.log 2>&1
cat $R/gpurun_out/tmp.log
