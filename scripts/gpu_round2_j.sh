#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r02_j
mkdir -p $out
timeout 1500 python3 -m pytest tests/test_distributed_gpu.py tests/test_tt06_known_answer.py -x -q -m gpu > $out/pytest.log 2>&1
tail -25 $out/pytest.log
