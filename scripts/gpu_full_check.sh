#!/bin/bash
# the driver's round-end sequence: GPU tests, smoke, bench
mkdir -p gpurun_out
{
timeout 3300 python3 -m pytest tests -q -m gpu -x 2>&1 | tail -25
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python3 bench.py 2>/dev/null | tail -1
} > gpurun_out/full_check.log 2>&1
tail -c 7000 gpurun_out/full_check.log
