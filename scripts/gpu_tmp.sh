#!/bin/bash
mkdir -p gpurun_out
{
for st in atomic color; do
  echo "== $st"
  timeout 300 python3 scripts/bench_mechanics.py --n 80 --steps 2 --cpu-n 2 --strategy $st 2>&1 | tail -1 | cut -c1-260
done
} > gpurun_out/tmp.log 2>&1
cat gpurun_out/tmp.log
