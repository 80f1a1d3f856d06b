#!/bin/bash
# Q2 scalar forms: wave roles by SIMD id; 4x4x4 (3 per CU) against the 16x16 form
mkdir -p gpurun_out
export TMPDIR=/tmp
one() {
python3 -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "q2_scalar or quadratic_field" --tb=short 2>&1 | tail -2
python3 scripts/bench_q2_scalar.py --n 64 --strategies element
rm -rf /tmp/q2prof; rocprofv3 --kernel-trace --stats -d /tmp/q2prof -o q2 -- python3 scripts/bench_q2_scalar.py --n 64 --strategies element > /dev/null 2>&1
db=$(find /tmp/q2prof -name "*.db" | head -1)
python3 scripts/rocpd_summary.py "$db" | head -3 | sed 's/EEvNS.*kd//'
}
{
echo "== 4x4, 3 per CU"; one
echo "== 16x16"; TB_Q2_KERNEL=16x16 one
} > gpurun_out/u.log 2>&1
tail -c 5000 gpurun_out/u.log
