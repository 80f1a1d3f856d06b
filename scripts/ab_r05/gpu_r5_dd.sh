#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r05_v2; mkdir -p $out
python3 examples/monodomain_fhn.py --ionic tt06 --n 216 --steps 40 > $out/monodomain_tt06_216.json 2>/dev/null
cd /tmp; rm -rf /tmp/kte; rocprofv3 --kernel-trace --stats -d /tmp/kte -o kte -- python3 $GRAFT_REPO_ROOT/examples/monodomain_fhn.py --ionic tt06 --n 216 > /tmp/mono_traced.json 2>/dev/null
cd $GRAFT_REPO_ROOT; python3 scripts/rocpd_summary.py /tmp/kte/kte_results.db --json $out/monodomain_kernel_stats.json > $out/monodomain_kernel_stats.txt
tail -1 $out/monodomain_tt06_216.json | cut -c1-400; tail -1 /tmp/mono_traced.json | cut -c1-300
bash scripts/gpu_check.sh
