#!/bin/bash
# Round 5: device-AD energies through the point kernel — parity, A/B at 40³ (profiling library: TB_MECH_SPLIT switch), graph test
mkdir -p gpurun_out/r5
R=$GRAFT_REPO_ROOT
{
python3 -m pytest tests/test_gpu_parity.py -q -x -k "graph_replay or other_energies or contracting_cuboid or land2015 or condensed or hyperelastic or prestress or hill or ideal_lv_load or config5" 2>&1 | tail -6
run() { echo "== $*"; env "$@" timeout 900 python3 scripts/bench_mechanics.py --n 40 --steps 3 --cpu-n 2 --energy guccione 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('  guccione 40^3 linearize %.2f ms residual %.2f ms' % (d['linearize_ms'], d['residual_ms']))"; }
export TB_LIBTBHIP=$R/thunderbolt.jl_amd/libtbhip_ablation.so
for rep in 1 2; do
  run TB_MECH_SPLIT=0
  run TB_MECH_SPLIT=1
done
for en in humphrey linyin; do
  for sp in 0 1; do echo "== $en split=$sp"; TB_MECH_SPLIT=$sp timeout 900 python3 scripts/bench_mechanics.py --n 40 --steps 3 --cpu-n 2 --energy $en 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('  linearize %.2f ms' % d['linearize_ms'])"; done
done
echo "== HO 40^3"; for sp in 0 1; do TB_MECH_SPLIT=$sp timeout 900 python3 scripts/bench_mechanics.py --n 40 --steps 3 --cpu-n 2 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('  split=$sp linearize %.2f ms' % d['linearize_ms'])"; done
unset TB_LIBTBHIP
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python3 $R/scripts/bench_mechanics.py --n 40 --steps 3 --cpu-n 2 --energy guccione > /tmp/kt.log 2>&1
python3 $R/scripts/rocpd_summary.py $(find /tmp/kt -name "*.db" | head -1) 2>&1 | cut -c1-200 | head -6
} > gpurun_out/r5/d.log 2>&1
tail -c 5000 $R/gpurun_out/r5/d.log
