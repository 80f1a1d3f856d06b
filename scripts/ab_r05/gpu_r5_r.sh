#!/bin/bash
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
{
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x -k "q2 or quadratic or Q2 or order2 or lagrange2" 2>&1 | tail -3
for i in 1 2; do
echo "tensor-order:"; python3 scripts/bench_q2_scalar.py --n 64 --strategies element 2>/dev/null | tail -1
echo "ferrite-order:"; TB_LIBTBHIP=$PWD/thunderbolt.jl_amd/libtbhip_q2old.so python3 scripts/bench_q2_scalar.py --n 64 --strategies element 2>/dev/null | tail -1
done
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/ktq -o ktq -- python3 $GRAFT_REPO_ROOT/scripts/bench_q2_scalar.py --n 64 --strategies element > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python3 scripts/rocpd_summary.py /tmp/ktq/ktq_results.db | head -5 | cut -c1-150
} > gpurun_out/r5/r.log 2>&1
cat gpurun_out/r5/r.log | grep -v amdgpu.ids
