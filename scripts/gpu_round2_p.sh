#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r02_p
mkdir -p $out
(time timeout 600 python3 examples/land2015_beam.py) 2>&1 | grep -v amdgpu | tail -6 | tee $out/land_cheb.txt
(time timeout 600 python3 examples/land2015_beam.py --inner cg) 2>&1 | grep -v amdgpu | tail -6 | tee $out/land_cg.txt
(time timeout 900 python3 examples/electromechanics_lv.py --inner gmres) 2>&1 | grep -v amdgpu | tail -8 | cut -c1-400 | tee $out/em_gmres.txt
(time timeout 900 python3 examples/electromechanics_lv.py --inner lu) 2>&1 | grep -v amdgpu | tail -4 | cut -c1-400 | tee $out/em_lu.txt
