#!/bin/bash
# Round 5, config 4: isolated kernel times of the split linearisation (one launch of each kernel), variants, chunk counts
mkdir -p gpurun_out/r5
R=$GRAFT_REPO_ROOT
{
run() { echo "== $*"; env "$@" timeout 600 python3 scripts/bench_mechanics.py --n 80 --steps 3 --cpu-n 2 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('  linearize %.2f ms residual %.2f ms' % (d['linearize_ms'], d['residual_ms']))"; }
for rep in 1 2; do
  run TB_MECH_SPLIT=0
  run TB_MECH_SPLIT=1
  run TB_MECH_SPLIT=1 TB_LIBTBHIP=$R/thunderbolt.jl_amd/libtbhip_mp1.so
  run TB_MECH_SPLIT=1 TB_LIBTBHIP=$R/thunderbolt.jl_amd/libtbhip_c3.so
  run TB_MECH_SPLIT=1 TB_MECH_CHUNKS=32
  run TB_MECH_SPLIT=1 TB_MECH_CHUNKS=32 TB_LIBTBHIP=$R/thunderbolt.jl_amd/libtbhip_c3.so
done
cd /tmp && export TMPDIR=/tmp
for v in "" mp1; do
  rm -rf /tmp/kt
  export TB_MECH_CHUNKS=0
  [ -n "$v" ] && export TB_LIBTBHIP=$R/thunderbolt.jl_amd/libtbhip_$v.so
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python3 $R/scripts/bench_mechanics.py --n 80 --steps 3 --cpu-n 2 > /tmp/kt.log 2>&1
  echo "== kernel trace, one launch per kernel (TB_MECH_CHUNKS=0) variant '$v'"
  python3 $R/scripts/rocpd_summary.py $(find /tmp/kt -name "*.db" | head -1) 2>&1 | cut -c1-200 | head -6
done
} > gpurun_out/r5/mech_ab2.log 2>&1
tail -c 5000 $R/gpurun_out/r5/mech_ab2.log
