#!/bin/bash
mkdir -p gpurun_out/r5
{
echo "== current"; python3 scripts/gmres_time.py 128
echo "== previous commit (single-address sums, 256-thread workgroups)"; TB_LIBTBHIP=$PWD/thunderbolt.jl_amd/libtbhip_prev.so python3 scripts/gmres_time.py 128
echo "== current"; python3 scripts/gmres_time.py 216
echo "== previous"; TB_LIBTBHIP=$PWD/thunderbolt.jl_amd/libtbhip_prev.so python3 scripts/gmres_time.py 216
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -q -x -k "gmres or newton or pcg or cheb or l1gs or solve or cg or heat" 2>&1 | tail -3
} > gpurun_out/r5/q.log 2>&1
cat gpurun_out/r5/q.log | grep -v amdgpu.ids
