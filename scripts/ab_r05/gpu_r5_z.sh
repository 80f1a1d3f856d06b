#!/bin/bash
mkdir -p gpurun_out/r5
{
t0=$SECONDS
TB_BENCH_SHARE_DEVICE=1 timeout 240 python3 bench.py --gpus 2 --n 24 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -3 | cut -c1-600
echo "rc $? wall $((SECONDS-t0)) s"
t0=$SECONDS
TB_BENCH_SHARE_DEVICE=1 timeout 240 python3 bench.py --gpus 2 --n 24 --steps 2 --warmup 1 --no-cpu-baseline --no-overlap-exchange 2>&1 | tail -1 | cut -c1-300
echo "rc $? wall $((SECONDS-t0)) s"
timeout 1200 python3 -m pytest tests/test_distributed_gpu.py tests/test_rccl_world1.py -q -x 2>&1 | tail -4
} > gpurun_out/r5/z.log 2>&1
grep -v amdgpu.ids gpurun_out/r5/z.log
