#!/bin/bash
mkdir -p gpurun_out
{
timeout 300 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x --timeout 200 -k "q2_scalar or quadratic_field" 2>&1 | tail -5
for mode in sf mfma; do
  echo "== $mode"
  if [ $mode = mfma ]; then export TB_Q2_KERNEL=mfma; else unset TB_Q2_KERNEL; fi
  timeout 300 python3 scripts/bench_q2_scalar.py --n 64 2>&1 | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.readline()); print({k:round(v['ms'],3) for k,v in d.items() if isinstance(v,dict)})"
done
} > gpurun_out/q2_check.log 2>&1
cat gpurun_out/q2_check.log | cut -c1-400
