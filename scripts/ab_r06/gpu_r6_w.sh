# W: kernel trace of the N-rank step on one GPU (27 layers, exchange through the C ABI in the device's queue, rank 0 its own two neighbours): who takes the 28 us
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out/r6w
TB_BENCH_SELF_EXCHANGE=abi rocprofv3 --kernel-trace -d gpurun_out/r6w/kt -o kt -- python3 bench.py --layers 27 --steps 40 --warmup 10 --no-slab-sweep --no-cpu-baseline --no-graph --no-dist-cg --trace-markers > gpurun_out/r6w/line.json 2> gpurun_out/r6w/err.txt
db=$(find gpurun_out/r6w/kt -name '*.db' | head -1)
python3 scripts/rocpd_timeline.py $db --between scan --skip 40 --count 36 > gpurun_out/r6w/timeline_abi.txt
python3 scripts/rocpd_summary.py $db --between scan > gpurun_out/r6w/stats_abi.txt
TB_BENCH_SELF_EXCHANGE=torch rocprofv3 --kernel-trace -d gpurun_out/r6w/kt2 -o kt -- python3 bench.py --layers 27 --steps 40 --warmup 10 --no-slab-sweep --no-cpu-baseline --no-graph --no-dist-cg --trace-markers > gpurun_out/r6w/line2.json 2>> gpurun_out/r6w/err.txt
db=$(find gpurun_out/r6w/kt2 -name '*.db' | head -1)
python3 scripts/rocpd_timeline.py $db --between scan --skip 40 --count 36 > gpurun_out/r6w/timeline_torch.txt
rm -rf gpurun_out/r6w/kt gpurun_out/r6w/kt2
cat gpurun_out/r6w/timeline_abi.txt; echo; cat gpurun_out/r6w/timeline_torch.txt; tail -3 gpurun_out/r6w/err.txt
