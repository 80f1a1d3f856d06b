#!/bin/bash
mkdir -p gpurun_out/r5
R=$GRAFT_REPO_ROOT
cd $R/scripts/microbench && hipcc --offload-arch=gfx950 -O3 partial_rows.hip -o partial_rows.bin 2>&1 | tail -3
{
./partial_rows.bin
cd /tmp && export TMPDIR=/tmp
for grp in "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  rm -rf /tmp/pr; timeout 300 rocprofv3 --pmc $grp -d /tmp/pr -o pmc -- $R/scripts/microbench/partial_rows.bin > /dev/null 2>&1
  python3 $R/scripts/rocpd_summary.py $(find /tmp/pr -name "*.db" | head -1) 2>&1 | cut -c1-140
done
} > $R/gpurun_out/r5/partial_rows.txt 2>&1
cat $R/gpurun_out/r5/partial_rows.txt
