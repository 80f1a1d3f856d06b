#!/bin/bash
mkdir -p gpurun_out/r5
{
python3 -m pytest tests/test_gpu_parity.py tests/test_rccl_world1.py tests/test_distributed_gpu.py -q -x -k "cgd_iteration or bisection or rccl or bench_two_ranks" 2>&1 | grep -v "^$" | tail -30 | cut -c1-250
} > gpurun_out/r5/g.log 2>&1
cat gpurun_out/r5/g.log
