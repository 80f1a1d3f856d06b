#!/bin/bash
# round 4, first GPU contact: the new tests, then the bench line with the slab sweep
mkdir -p gpurun_out
{
timeout 900 python3 -m pytest tests/test_rccl_world1.py tests/test_distributed_gpu.py -q -x 2>&1 | tail -15
timeout 600 python3 -m pytest tests/test_gpu_parity.py -q -x -k "element_strategy or matrix_parity or source_vector" 2>&1 | tail -5
timeout 900 python3 bench.py 2> gpurun_out/bench_first.err | tail -1 > gpurun_out/bench_first.json
tail -c 3000 gpurun_out/bench_first.json
tail -5 gpurun_out/bench_first.err
} > gpurun_out/gpu_r4_first.log 2>&1
tail -c 7000 gpurun_out/gpu_r4_first.log
