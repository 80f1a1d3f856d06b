#!/bin/bash
# targeted GPU tests: $1 = pytest -k expression
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests -q -m gpu --timeout 300 -x -k "$1" 2>&1 | tail -30 > gpurun_out/quick.log
cat gpurun_out/quick.log | cut -c1-400
