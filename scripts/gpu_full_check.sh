#!/bin/bash
# the driver's round-end sequence: GPU tests, smoke, bench (each under its own timeout); "$1" = extra pytest options (e.g. none → no -x)
mkdir -p gpurun_out
{
timeout 1200 python3 -m pytest tests -q -m gpu --timeout 300 $1 2>&1 | tail -40
timeout 120 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 500 python3 bench.py 2>/dev/null | tail -1
} > gpurun_out/full_check.log 2>&1
tail -c 9000 gpurun_out/full_check.log
