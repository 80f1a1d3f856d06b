#!/bin/bash
# round 6, experiment M: config 4's contraction with the three row components side by side and one lane per three column nodes (k_mech_contract3)
# against round 5's k_mech_contract (TB_MECH_CONTRACT=c1, profiling build)
mkdir -p gpurun_out
L=$PWD/thunderbolt.jl_amd/libtbhip_ablation.so
{
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "hyperelastic or mechanics" 2>&1 | tail -3
for rep in 1 2; do
  for v in c3 c1; do
    echo "== $v"
    TB_LIBTBHIP=$L TB_MECH_CONTRACT=$v timeout 900 python3 scripts/bench_mechanics.py --n 80 --cpu-n 2 --steps 5 2>&1 | tail -1 | cut -c1-420
  done
done
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/ktm6 -o ktm -- python3 scripts/bench_mechanics.py --n 80 --cpu-n 2 --steps 3 > /dev/null 2>&1
python3 scripts/rocpd_summary.py gpurun_out/ktm6/ktm_results.db 2>/dev/null | head -8 | cut -c1-170
rm -rf gpurun_out/ktm6
} > gpurun_out/gpu_r6_m.log 2>&1
cat gpurun_out/gpu_r6_m.log
