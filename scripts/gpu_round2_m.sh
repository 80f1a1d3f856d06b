#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r02_m
mkdir -p $out
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "hyperelastic or mechanics or subdomains or contracting or land2015 or condensed" > $out/pytest.log 2>&1
tail -4 $out/pytest.log
python3 scripts/bench_mechanics.py --n 80 --cpu-n 2 2>/dev/null | cut -c1-330 | tee $out/mech_src_gather.json
TB_EA_GATHER=positions python3 scripts/bench_mechanics.py --n 80 --cpu-n 2 2>/dev/null | cut -c1-330 | tee $out/mech_pos_gather.json
rocprofv3 --kernel-trace --stats -d $out/ktm -o ktm -- python3 scripts/bench_mechanics.py --n 80 --cpu-n 2 > /dev/null 2>&1
python3 scripts/rocpd_summary.py $out/ktm/ktm_results.db --json $out/mechanics_kernel_stats.json | head -6 | cut -c1-200
rm -rf $out/ktm
