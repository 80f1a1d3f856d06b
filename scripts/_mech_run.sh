cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu -k "hyperelastic or mechanic or newton or weak_boundary or active" 2>&1 | tail -3
for st in element atomic color; do python3 scripts/bench_mechanics.py --n 40 --cpu-n 2 --strategy $st 2>&1 | tail -1 | cut -c1-300; done
python3 scripts/bench_mechanics.py --n 80 --cpu-n 2 2>&1 | tail -1 | cut -c1-300
