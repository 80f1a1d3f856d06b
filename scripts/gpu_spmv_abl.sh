#!/bin/bash
# Stream SpMV, profiling build: the product with and without the gather of x (TB_SPMV_NOGATHER: coalesced reads of the same volume instead), 216³
mkdir -p gpurun_out
{
export TB_LIBTBHIP=$PWD/thunderbolt.jl_amd/libtbhip_ablation.so
for rep in 1 2; do
echo "== with gather"; timeout 300 python3 scripts/bench_spmv.py --n 216 2>&1 | grep "spmv" | cut -c1-200
echo "== no gather";  TB_SPMV_NOGATHER=1 timeout 300 python3 scripts/bench_spmv.py --n 216 2>&1 | grep "spmv" | cut -c1-200
done
} > gpurun_out/spmv_abl.log 2>&1
cat gpurun_out/spmv_abl.log
