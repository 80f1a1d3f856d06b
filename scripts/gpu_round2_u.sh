#!/bin/bash
# Q2 scalar forms on the matrix cores: parity, timing of the three strategies, per-kernel split of the element strategy
mkdir -p gpurun_out
export TMPDIR=/tmp
{
python3 -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "q2_scalar or quadratic_field or strategies" --tb=short 2>&1 | tail -3
python3 scripts/bench_q2_scalar.py --n 64
rm -rf /tmp/q2prof; rocprofv3 --kernel-trace --stats -d /tmp/q2prof -o q2 -- python3 scripts/bench_q2_scalar.py --n 64 --strategies element > /dev/null 2>&1
db=$(find /tmp/q2prof -name "*.db" | head -1)
python3 scripts/rocpd_summary.py "$db" | head -4
} > gpurun_out/u.log 2>&1
tail -c 5000 gpurun_out/u.log
