#!/bin/bash
# Run on the GPU box (gpurun -- 'bash scripts/collect_profiles.sh <tag>'): rocprofv3 evidence for profiles/<tag>/.
# Kernel-trace and each PMC group are separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950).
tag=${1:-r01_v4}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/$tag
mkdir -p "$out"
python3 bench.py > "$out/bench_216.json" 2> "$out/bench_216.err"
rocprofv3 --kernel-trace --stats -d "$out/kt" -o kt -- python3 bench.py --no-cpu-baseline > "$out/bench_under_rocprof.json" 2>/dev/null
python3 scripts/rocpd_summary.py "$out/kt/kt_results.db" --json "$out/kernel_stats.json" > "$out/kernel_stats.txt"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d "$out/pmc$i" -o pmc -- python3 bench.py --no-cpu-baseline > /dev/null 2>&1
  python3 scripts/rocpd_summary.py "$out/pmc$i/pmc_results.db" --json "$out/pmc$i.json" > /dev/null
done
# mechanics (BASELINE config 4)
python3 scripts/bench_mechanics.py --n 80 > "$out/mechanics_80_q2.json" 2>/dev/null
rocprofv3 --kernel-trace --stats -d "$out/ktm" -o ktm -- python3 scripts/bench_mechanics.py --n 80 --cpu-n 2 > /dev/null 2>&1
python3 scripts/rocpd_summary.py "$out/ktm/ktm_results.db" --json "$out/mechanics_kernel_stats.json" > "$out/mechanics_kernel_stats.txt"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d "$out/pmcm" -o pmcm -- python3 scripts/bench_mechanics.py --n 40 --cpu-n 2 > /dev/null 2>&1
python3 scripts/rocpd_summary.py "$out/pmcm/pmcm_results.db" --json "$out/mechanics_pmc_40.json" > /dev/null
# monodomain simulation (config 3 as a time loop): the stream SpMV + CG kernels next to the reaction step
rocprofv3 --kernel-trace --stats -d "$out/kte" -o kte -- python3 examples/monodomain_fhn.py --ionic tt06 --n 216 > "$out/monodomain_tt06_216.json" 2>/dev/null
python3 scripts/rocpd_summary.py "$out/kte/kte_results.db" --json "$out/monodomain_kernel_stats.json" > "$out/monodomain_kernel_stats.txt"
rm -rf "$out"/kte
rm -rf "$out"/kt "$out"/pmc1 "$out"/pmc2 "$out"/pmc3 "$out"/ktm "$out"/pmcm
./scripts/microbench/mfma_f64.bin > "$out/mfma_f64_microbench.txt" 2>&1
tail -c 600 "$out/bench_216.json"; cat "$out/mechanics_80_q2.json" | cut -c1-300; head -5 "$out/kernel_stats.txt"
