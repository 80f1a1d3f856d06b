#!/bin/bash
# One gpurun call that replays what the driver does at round end, in its order: the GPU tests, smoke(), the bench line.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- bash scripts/gpu_check.sh
mkdir -p gpurun_out
{
python3 -m pytest tests -q -m gpu -x 2>&1 | tail -4
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python3 bench.py 2>/dev/null | tail -1
} > gpurun_out/gpu_check.log 2>&1
tail -c 6000 gpurun_out/gpu_check.log
