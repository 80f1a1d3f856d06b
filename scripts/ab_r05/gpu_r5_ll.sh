#!/bin/bash
out=gpurun_out/r05_v2; mkdir -p $out
python3 scripts/bench_tets.py --n 100 > $out/tets_100.json 2>/dev/null
python3 scripts/bench_spmv.py > $out/spmv_216.json 2>/dev/null
python3 scripts/bench_reaction.py > $out/reaction_models.json 2>/dev/null
cut -c1-700 $out/tets_100.json; cut -c1-500 $out/spmv_216.json; cat $out/reaction_models.json
