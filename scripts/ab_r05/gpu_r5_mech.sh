#!/bin/bash
# Round 5, config 4: parity of the split linearisation + A/B against the fused kernel at 40³ and 80³ (one gpurun call)
mkdir -p gpurun_out/r5
{
python3 -m pytest tests/test_gpu_parity.py -q -x -k "hyperelastic or chunked_linearization or condensed_sarcomere_parity or nodal_fibre" 2>&1 | tail -5
for rep in 1 2; do
  for envs in "TB_MECH_SPLIT=0" "TB_MECH_SPLIT=1" "TB_MECH_SPLIT=1 TB_MECH_CHUNKS=0" "TB_MECH_SPLIT=0 TB_MECH_CHUNKS=0" "TB_MECH_SPLIT=1 TB_MECH_CHUNKS=16"; do
    echo "== $envs"
    env $envs timeout 600 python3 scripts/bench_mechanics.py --n 80 --steps 3 --cpu-n 2 2>&1 | tail -1 | cut -c1-330
  done
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python3 $GRAFT_REPO_ROOT/scripts/bench_mechanics.py --n 80 --steps 3 --cpu-n 2 > /tmp/kt.log 2>&1
python3 $GRAFT_REPO_ROOT/scripts/rocpd_summary.py $(find /tmp/kt -name "*.db" | head -1) 2>&1 | cut -c1-200 | head -12
} > gpurun_out/r5/mech_ab.log 2>&1
tail -c 5000 gpurun_out/r5/mech_ab.log
