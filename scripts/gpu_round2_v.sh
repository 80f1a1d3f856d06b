#!/bin/bash
# mechanics gather with three component rows per pass: parity + timing split at 80^3
mkdir -p gpurun_out
export TMPDIR=/tmp
{
python3 -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "hyperelastic" --tb=short 2>&1 | tail -3
python3 scripts/bench_mechanics.py --n 80 --cpu-n 2 --steps 3
rm -rf /tmp/mprof; rocprofv3 --kernel-trace --stats -d /tmp/mprof -o m -- python3 scripts/bench_mechanics.py --n 80 --cpu-n 2 --steps 3 > /dev/null 2>&1
db=$(find /tmp/mprof -name "*.db" | head -1)
python3 scripts/rocpd_summary.py "$db" | head -5 | cut -c1-200
} > gpurun_out/v.log 2>&1
tail -c 4000 gpurun_out/v.log
