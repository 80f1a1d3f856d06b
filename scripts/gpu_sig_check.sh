#!/bin/bash
# Fused M + K patch kernel: signature table in LDS (default) against read from global memory — parity tests, then bench at 216³ three times each
mkdir -p gpurun_out
{
timeout 300 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "patch or properties_100 or strategies" 2>&1 | tail -3
if [ ${PIPESTATUS[0]} -ne 0 ]; then echo "parity failed or hung: stop"; exit 1; fi
for rep in 1 2 3; do
for mode in lds global; do
  if [ $mode = global ]; then export TB_PATCH_SIG=global; else unset TB_PATCH_SIG; fi
  echo "== $mode"
  timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dist-cg 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'])" 
done
done
} > gpurun_out/sig_check.log 2>&1
tail -c 3000 gpurun_out/sig_check.log
