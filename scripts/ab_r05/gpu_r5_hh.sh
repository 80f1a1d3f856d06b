#!/bin/bash
mkdir -p gpurun_out/r5
{ timeout 600 python3 -m pytest tests/test_gpu_parity.py -q -x -k "reduction_slots or cgd_iteration" 2>&1 | grep -v amdgpu | tail -15; } > gpurun_out/r5/hh.log 2>&1
cat gpurun_out/r5/hh.log
