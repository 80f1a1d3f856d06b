#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r02_q
mkdir -p $out
timeout 1500 python3 -m pytest tests/test_examples.py tests/test_gpu_parity.py -x -q -m gpu -k "examples or l1_gauss or error_codes" > $out/pytest.log 2>&1
tail -5 $out/pytest.log
