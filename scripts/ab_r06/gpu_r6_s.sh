# S: cost of a fork / join between two queues for the forking queue (scripts/microbench/cross_queue.py), filler ≈ 100 µs and ≈ 20 µs
mkdir -p gpurun_out/r6s
python3 scripts/microbench/cross_queue.py 24000000 2> gpurun_out/r6s/err.txt | grep '^{' > gpurun_out/r6s/cross_queue_100us.json
python3 scripts/microbench/cross_queue.py 3000000 2>> gpurun_out/r6s/err.txt | grep '^{' > gpurun_out/r6s/cross_queue_20us.json
cat gpurun_out/r6s/*.json; tail -5 gpurun_out/r6s/err.txt
