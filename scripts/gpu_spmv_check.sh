#!/bin/bash
# Stream SpMV: record-driven kernel (default) against the five-trip kernel (TB_SPMV_KERNEL=chain) — parity tests, then timing at 216³ and 100³
mkdir -p gpurun_out
{
timeout 600 python3 -m pytest tests/test_gpu_parity.py tests/test_distributed_gpu.py -q -m gpu -x --timeout 300 -k "spmv or cg or heat or halo or monodomain or distributed" 2>&1 | tail -3
if [ ${PIPESTATUS[0]} -ne 0 ]; then echo "parity failed or hung: stop"; exit 1; fi
for rep in 1 2; do
for k in rows rec; do
  echo "== $k"
  TB_SPMV_KERNEL=$k timeout 300 python3 scripts/bench_spmv.py --n 216 2>&1 | grep "spmv\|rel err\|cg" | cut -c1-200
done
done
TB_SPMV_KERNEL=rows timeout 300 python3 scripts/bench_spmv.py --n 100 2>&1 | grep "spmv" | cut -c1-200
TB_SPMV_KERNEL=rec timeout 300 python3 scripts/bench_spmv.py --n 100 2>&1 | grep "spmv" | cut -c1-200
} > gpurun_out/spmv_check.log 2>&1
cat gpurun_out/spmv_check.log
