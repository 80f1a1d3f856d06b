#!/bin/bash
mkdir -p gpurun_out/r5
R=$GRAFT_REPO_ROOT
{
python3 -m pytest tests/test_gpu_parity.py -q -x -k "other_energies or contracting_cuboid or land2015 or prestress or hill" 2>&1 | tail -4
for en in guccione humphrey linyin ho; do
  echo "== $en"; timeout 900 python3 scripts/bench_mechanics.py --n 40 --steps 3 --cpu-n 2 --energy $en 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('  linearize %.2f ms residual %.2f' % (d['linearize_ms'], d['residual_ms']))"
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python3 $R/scripts/bench_mechanics.py --n 40 --steps 3 --cpu-n 2 --energy guccione > /tmp/kt.log 2>&1
python3 $R/scripts/rocpd_summary.py $(find /tmp/kt -name "*.db" | head -1) 2>&1 | cut -c1-200 | head -5
} > gpurun_out/r5/e.log 2>&1
cat gpurun_out/r5/e.log
