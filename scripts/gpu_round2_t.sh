#!/bin/bash
# baseline numbers for the Q2 paths before reworking them
mkdir -p gpurun_out
{
python3 scripts/bench_q2_scalar.py --n 64
for s in atomic color element; do python3 scripts/bench_mechanics.py --n 40 --strategy $s --cpu-n 4 --steps 3; done
python3 scripts/bench_mechanics.py --n 80 --strategy atomic --cpu-n 4 --steps 2
} > gpurun_out/t.log 2>&1
tail -c 6000 gpurun_out/t.log
