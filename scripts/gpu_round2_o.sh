#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r02_o
mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "l1_gauss_seidel or error_codes" > $out/pytest.log 2>&1
tail -8 $out/pytest.log
python3 scripts/bench_precond.py --n 100 --mech-n 24 2>/dev/null | tee $out/precond.json
