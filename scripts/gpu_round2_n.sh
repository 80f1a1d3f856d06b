#!/bin/bash
# the driver's round-end sequence: GPU test tier, smoke, default bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r02_n
mkdir -p $out
( time timeout 2400 python3 -m pytest tests/ -x -q -m gpu ) > $out/pytest_gpu.log 2>&1
tail -6 $out/pytest_gpu.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -3
( time python3 bench.py ) > $out/bench_default.json 2> $out/bench_default.err
tail -c 900 $out/bench_default.json; tail -4 $out/bench_default.err
