#!/bin/bash
# ablation + counters of the sum-factorised patch kernels
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r02_b
mkdir -p $out
for f in 0 1 2 4 3 5 6 7; do
  TB_LIBTBHIP=$PWD/thunderbolt.jl_amd/libtbhip_ablation.so TB_DEBUG_FLAGS=$f python3 scripts/bench_assembly.py --n 216 2>/dev/null | sed "s/^/flags=$f /" | tee -a $out/ablation.txt
done
rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- python3 scripts/bench_assembly.py --n 216 > /dev/null 2>&1
python3 scripts/rocpd_summary.py $out/kt/kt_results.db --json $out/kernel_stats.json > $out/kernel_stats.txt; head -8 $out/kernel_stats.txt
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d $out/pmc1 -o pmc -- python3 scripts/bench_assembly.py --n 216 --reps 3 > /dev/null 2>&1
python3 scripts/rocpd_summary.py $out/pmc1/pmc_results.db --json $out/pmc_sq.json | head -40
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM -d $out/pmc2 -o pmc -- python3 scripts/bench_assembly.py --n 216 --reps 3 > /dev/null 2>&1
python3 scripts/rocpd_summary.py $out/pmc2/pmc_results.db --json $out/pmc_sq2.json | head -40
rm -rf $out/kt $out/pmc1 $out/pmc2
