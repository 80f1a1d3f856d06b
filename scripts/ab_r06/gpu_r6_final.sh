set -x
mkdir -p gpurun_out/final_r6
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/final_r6/pytest_gpu.txt
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_r6/smoke.txt 2>&1
python3 bench.py > gpurun_out/final_r6/bench.json 2> gpurun_out/final_r6/bench.err
tail -3 gpurun_out/final_r6/pytest_gpu.txt; tail -2 gpurun_out/final_r6/smoke.txt; cut -c1-600 gpurun_out/final_r6/bench.json
