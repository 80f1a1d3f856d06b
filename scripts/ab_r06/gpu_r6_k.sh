#!/bin/bash
# round 6, experiment K: config 4 (80^3 Q2 Holzapfel-Ogden tangent) with chunks small enough for a chunk's element matrices to stay in the 256 MB
# memory-side cache between the contraction that writes them and the gather that reads them (52.5 KB per cell: 128 chunks = 210 MB, 256 = 105 MB),
# lexicographic and Morton cell order (compact chunks: fewer rows wait for a later chunk)
mkdir -p gpurun_out
{
for order in lexicographic morton; do
  echo "== $order"
  timeout 1500 python3 scripts/bench_mechanics.py --n 80 --cell-order $order --cpu-n 2 --steps 3 --chunks 1,8,32,64,128,256,512 2>&1 | tail -1 | cut -c1-1200
done
} > gpurun_out/gpu_r6_k.log 2>&1
cat gpurun_out/gpu_r6_k.log
