#!/bin/bash
mkdir -p gpurun_out/r5
{
python3 -m pytest tests/test_gpu_parity.py -q -x -k "graph_replay" 2>&1 | grep -v "^$" | tail -40 | cut -c1-250
echo "== rccl latency probe"
timeout 200 python3 scripts/rccl_latency.py 47089; echo "rc=$?"
} > gpurun_out/r5/c.log 2>&1
tail -c 6000 gpurun_out/r5/c.log
