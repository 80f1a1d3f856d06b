#!/bin/bash
# kernel times of the mechanics tangent, symmetric-packed vs full storage of the element matrices
mkdir -p gpurun_out/mechprof
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in sym full; do
  if [ $mode = full ]; then export TB_MECH_KE=full; else unset TB_MECH_KE; fi
  rm -rf /tmp/mp_$mode
  rocprofv3 --kernel-trace --stats -d /tmp/mp_$mode -o mech -- python3 $R/scripts/bench_mechanics.py --n 80 --steps 3 --cpu-n 2 > /tmp/mp_$mode.log 2>&1
  db=$(find /tmp/mp_$mode -name "*.db" | head -1)
  echo "== $mode ($db)" >> $R/gpurun_out/mechprof/summary.txt
  tail -1 /tmp/mp_$mode.log >> $R/gpurun_out/mechprof/summary.txt
  python3 $R/scripts/rocpd_summary.py $db $R/gpurun_out/mechprof/mech_${mode}_kernel_stats.json >> $R/gpurun_out/mechprof/summary.txt 2>&1
  find /tmp/mp_$mode -name "*stats*.csv" | head -3 >> $R/gpurun_out/mechprof/summary.txt
done
cat $R/gpurun_out/mechprof/summary.txt | cut -c1-220 | tail -40
