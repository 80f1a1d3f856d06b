#!/bin/bash
# SQ counters of the mechanics integration kernel (sum-factorised contraction), 40³
mkdir -p gpurun_out/mechprof
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/pm$i
  timeout 300 rocprofv3 --pmc $grp -d /tmp/pm$i -o pmc -- python3 $R/scripts/bench_mechanics.py --n 40 --steps 2 --cpu-n 2 > /tmp/pm$i.log 2>&1
  python3 $R/scripts/rocpd_summary.py $(find /tmp/pm$i -name "*.db" | head -1) --json $R/gpurun_out/mechprof/mech_sf_pmc$i.json 2>&1 | grep -A9 "k_hyperelastic.*Lb1ELb1ELi2\|k_gather_node_rows\|k_mech_" | cut -c1-150
done > $R/gpurun_out/mechprof/mech_sf_pmc.txt 2>&1
cat $R/gpurun_out/mechprof/mech_sf_pmc.txt
