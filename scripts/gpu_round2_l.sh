#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r02_l
mkdir -p $out
timeout 2400 python3 -m pytest tests -q -m gpu > $out/pytest_all.log 2>&1
grep -E "^FAILED|passed|failed|linear solve stopped" $out/pytest_all.log | tail -30
