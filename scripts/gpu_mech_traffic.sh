#!/bin/bash
# HBM traffic of the mechanics tangent kernels at 80³ (TCC request counters and WRITE_SIZE in separate passes)
mkdir -p gpurun_out/mechprof
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/tr$i
  timeout 400 rocprofv3 --pmc $grp -d /tmp/tr$i -o pmc -- python3 $R/scripts/bench_mechanics.py --n 80 --steps 2 --cpu-n 2 > /tmp/tr$i.log 2>&1
  python3 $R/scripts/rocpd_summary.py $(find /tmp/tr$i -name "*.db" | head -1) 2>&1 | grep -A5 "k_hyperelastic.*Lb1ELb1ELi2\|k_gather_node_rows\|k_mech_" | cut -c1-150
done > $R/gpurun_out/mechprof/mech_traffic_80.txt 2>&1
cat $R/gpurun_out/mechprof/mech_traffic_80.txt
