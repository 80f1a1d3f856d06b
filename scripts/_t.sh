cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu -k "heat or splitting or newton or tangent_fused" 2>&1 | tail -3
python3 examples/monodomain_fhn.py 2>&1 | tail -1 | cut -c1-400
python3 examples/mechanics_contraction.py 2>&1 | tail -1 | cut -c1-200
