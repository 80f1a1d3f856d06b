#!/bin/bash
# Mechanics tangent at 80³, profiling build: phase time stamps of every 256th workgroup (TB_PROF_STAMPS)
mkdir -p gpurun_out
{
export TB_LIBTBHIP=$PWD/thunderbolt.jl_amd/libtbhip_ablation.so TB_PROF_STAMPS=1
timeout 300 python3 scripts/bench_mechanics.py --n 80 --steps 2 --cpu-n 2 2>&1 | grep "mechanics phases\|linearize_ms" | cut -c1-400 | tail -4
} > gpurun_out/mech_stamps.log 2>&1
cat gpurun_out/mech_stamps.log
