#!/bin/bash
mkdir -p gpurun_out/r5
{
python3 examples/monodomain_fhn.py --ionic tt06 --n 216 --steps 30 --per-step 2>&1 | grep -v amdgpu | cut -c1-1500
} > gpurun_out/r5/bb.log 2>&1
cat gpurun_out/r5/bb.log
