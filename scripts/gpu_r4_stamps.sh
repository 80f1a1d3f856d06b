#!/bin/bash
# phase stamps of the staged and the record patch kernel (profiling build)
mkdir -p gpurun_out
{
for k in staged record; do
  echo "== $k"
  TB_LIBTBHIP=$PWD/thunderbolt.jl_amd/libtbhip_ablation.so TB_PROF_STAMPS=1 TB_PATCH_KERNEL=$k TB_PATCH_PREFETCH=0 timeout 600 python3 bench.py --no-cpu-baseline --no-slab-sweep --no-dist-cg --steps 3 --warmup 1 2>&1 | grep -E "staged phases" | tail -2
done
} > gpurun_out/gpu_r4_stamps.log 2>&1
cat gpurun_out/gpu_r4_stamps.log
