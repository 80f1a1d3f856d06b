# V: scalar Q2 element matrices stored through LDS (one run of 3 x 729 doubles per triple) against stored from the computing lanes (TB_Q2_STORE=direct, profiling build)
mkdir -p gpurun_out/r6v; rm -f gpurun_out/r6v/log.txt
python3 -m pytest tests -m gpu -q -k "q2 or Q2 or quadratic" 2>&1 | tail -3 >> gpurun_out/r6v/log.txt
export TB_LIBTBHIP=$PWD/thunderbolt.jl_amd/libtbhip_ablation.so
for rep in 1 2; do
  echo "== through LDS" >> gpurun_out/r6v/log.txt
  python3 scripts/bench_q2_scalar.py --n 64 --strategies element --steps 20 2>> gpurun_out/r6v/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print({k: round(v['ms'],4) for k,v in d.items() if isinstance(v,dict)})" >> gpurun_out/r6v/log.txt
  echo "== direct" >> gpurun_out/r6v/log.txt
  TB_Q2_STORE=direct python3 scripts/bench_q2_scalar.py --n 64 --strategies element --steps 20 2>> gpurun_out/r6v/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print({k: round(v['ms'],4) for k,v in d.items() if isinstance(v,dict)})" >> gpurun_out/r6v/log.txt
done
cat gpurun_out/r6v/log.txt; grep -v "amdgpu.ids" gpurun_out/r6v/err.txt | tail -5
