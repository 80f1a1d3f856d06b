#!/bin/bash
mkdir -p gpurun_out
{
timeout 1500 python3 -m pytest tests/test_rccl_world1.py tests/test_distributed_gpu.py -q -x 2>&1 | tail -15
} > gpurun_out/gpu_r4_rccl.log 2>&1
tail -c 5000 gpurun_out/gpu_r4_rccl.log
