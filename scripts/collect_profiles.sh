#!/bin/bash
# Run on the GPU box (gpurun -- 'bash scripts/collect_profiles.sh <tag>'): rocprofv3 evidence for profiles/<tag>/ and profiles/traffic.json.
# Kernel trace and every PMC group are separate passes (no PMC together with tracing domains; TCC read- and write-side counters do not fit one pass).
tag=${1:-r02_v1}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/$tag
mkdir -p "$out"
# micro-benchmarks are built here (binaries are not tracked)
for mb in mfma_f64 mfma_f64_4x4 lds_atomic wg_launch mall_reuse sell_spmv partial_rows tail_atomics; do
  [ -f scripts/microbench/$mb.hip ] && hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics scripts/microbench/$mb.hip -o scripts/microbench/$mb.bin 2>/dev/null
done
python3 bench.py > "$out/bench_216.json" 2> "$out/bench_216.err"
rocprofv3 --kernel-trace --stats -d "$out/kt" -o kt -- python3 bench.py --no-cpu-baseline --no-slab-sweep > "$out/bench_under_rocprof.json" 2>/dev/null
python3 scripts/rocpd_summary.py "$out/kt/kt_results.db" --json "$out/kernel_stats.json" > "$out/kernel_stats.txt"
i=0
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum" "WRITE_SIZE" "FETCH_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d "$out/pmc$i" -o pmc -- python3 bench.py --no-cpu-baseline --no-slab-sweep --steps 5 > /dev/null 2>&1
  python3 scripts/rocpd_summary.py "$out/pmc$i/pmc_results.db" --json "$out/pmc$i.json" > /dev/null
done
python3 scripts/make_traffic_json.py "$out" "$tag" > "$out/traffic.json"
if [ "$2" != "bench-only" ]; then
  # mechanics (BASELINE config 4)
  python3 scripts/bench_mechanics.py --n 80 > "$out/mechanics_80_q2.json" 2>/dev/null
  rocprofv3 --kernel-trace --stats -d "$out/ktm" -o ktm -- python3 scripts/bench_mechanics.py --n 80 --cpu-n 2 > /dev/null 2>&1
  python3 scripts/rocpd_summary.py "$out/ktm/ktm_results.db" --json "$out/mechanics_kernel_stats.json" > "$out/mechanics_kernel_stats.txt"
  # monodomain simulation (config 3 as a time loop)
  rocprofv3 --kernel-trace --stats -d "$out/kte" -o kte -- python3 examples/monodomain_fhn.py --ionic tt06 --n 216 > "$out/monodomain_tt06_216.json" 2>/dev/null
  python3 scripts/rocpd_summary.py "$out/kte/kte_results.db" --json "$out/monodomain_kernel_stats.json" > "$out/monodomain_kernel_stats.txt"
  [ -x scripts/microbench/mfma_f64.bin ] && ./scripts/microbench/mfma_f64.bin > "$out/mfma_f64_microbench.txt" 2>&1
  [ -x scripts/microbench/wg_launch.bin ] && ./scripts/microbench/wg_launch.bin > "$out/wg_launch_microbench.txt" 2>&1
  [ -x scripts/microbench/lds_atomic.bin ] && ./scripts/microbench/lds_atomic.bin > "$out/lds_atomic_microbench.txt" 2>&1
  [ -x scripts/microbench/mfma_f64_4x4.bin ] && ./scripts/microbench/mfma_f64_4x4.bin > "$out/mfma_f64_4x4_microbench.txt" 2>&1
  [ -x scripts/microbench/partial_rows.bin ] && ./scripts/microbench/partial_rows.bin > "$out/partial_rows_microbench.txt" 2>&1
  [ -x scripts/microbench/tail_atomics.bin ] && ./scripts/microbench/tail_atomics.bin > "$out/tail_atomics_microbench.txt" 2>&1
  python3 scripts/gmres_time.py 128 > "$out/krylov_solves_128.txt" 2>/dev/null
  python3 scripts/bench_tets.py --n 100 > "$out/tets_100.json" 2>/dev/null
  # coupled electromechanics on the ideal LV (config 5 as two kernel rates)
  python3 scripts/bench_electromechanics.py > "$out/electromechanics_lv.json" 2>/dev/null
  # the 27-layer slab of the strong-scaling split, kernel trace
  rocprofv3 --kernel-trace --stats -d "$out/kt27" -o kt27 -- python3 bench.py --no-cpu-baseline --layers 27 --steps 50 > /dev/null 2>&1
  python3 scripts/rocpd_summary.py "$out/kt27/kt27_results.db" --json "$out/slab27_kernel_stats.json" > "$out/slab27_kernel_stats.txt"
  # the CG product: CSR kernel against the sliced mirror, and the synthetic all-uniform-slices bound
  python3 scripts/bench_spmv.py > "$out/spmv_216.json" 2>/dev/null
  [ -x scripts/microbench/sell_spmv.bin ] && ./scripts/microbench/sell_spmv.bin > "$out/sell_spmv_microbench.txt" 2>&1
  # chunked mechanics linearisation: do integration and gather of neighbouring chunks run side by side? (kernel timeline of the last linearisations)
  rocprofv3 --kernel-trace --output-format csv -d "$out/ktc" -- python3 scripts/bench_mechanics.py --n 80 --steps 2 --cpu-n 2 > /dev/null 2>&1
  python3 scripts/trace_timeline.py "$out/ktc" "" 44 > "$out/mechanics_chunk_timeline.txt" 2>&1
  rm -rf "$out/ktc"
  # round 5: unstructured presentations of the bench workload (random renumbering of the box; idealised ventricle), device-AD energies through the
  # split linearisation, RCCL cost at world size 1, HBM traffic and SQ counters of the split mechanics kernels
  python3 bench.py --mesh shuffled > "$out/bench_shuffled.json" 2> "$out/bench_shuffled.err"
  TB_PLAN_VERBOSE=1 python3 bench.py --mesh lv > "$out/bench_lv.json" 2> "$out/bench_lv.err"
  for en in guccione humphrey linyin ho; do python3 scripts/bench_mechanics.py --n 40 --energy $en --cpu-n 2 2>/dev/null | tail -1; done > "$out/mechanics_energies_40.json"
  python3 scripts/rccl_latency.py 47089 2>/dev/null | grep "^{" > "$out/rccl_latency_world1.json"
  for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum" "WRITE_SIZE"; do
    rm -rf /tmp/trm; rocprofv3 --pmc $grp -d /tmp/trm -o pmc -- python3 scripts/bench_mechanics.py --n 80 --steps 2 --cpu-n 2 > /dev/null 2>&1
    python3 scripts/rocpd_summary.py $(find /tmp/trm -name "*.db" | head -1) 2>&1 | grep -A4 "k_mech_\|k_gather_node_rows" | cut -c1-150
  done > "$out/mechanics_traffic_80.txt" 2>&1
  for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
    rm -rf /tmp/trm; rocprofv3 --pmc $grp -d /tmp/trm -o pmc -- python3 scripts/bench_mechanics.py --n 40 --steps 2 --cpu-n 2 > /dev/null 2>&1
    python3 scripts/rocpd_summary.py $(find /tmp/trm -name "*.db" | head -1) 2>&1 | grep -A9 "k_mech_\|k_gather_node_rows" | cut -c1-150
  done > "$out/mechanics_pmc_40.txt" 2>&1
  # scalar forms on the quadratic field
  python3 scripts/bench_q2_scalar.py --n 64 > "$out/q2_scalar_64.json" 2>/dev/null
  rocprofv3 --kernel-trace --stats -d "$out/ktq" -o ktq -- python3 scripts/bench_q2_scalar.py --n 64 --strategies element > /dev/null 2>&1
  python3 scripts/rocpd_summary.py "$out/ktq/ktq_results.db" --json "$out/q2_scalar_kernel_stats.json" > "$out/q2_scalar_kernel_stats.txt"
fi
rm -rf "$out"/kt27 "$out"/kt "$out"/pmc1 "$out"/pmc2 "$out"/pmc3 "$out"/pmc4 "$out"/ktm "$out"/kte "$out"/ktq
tail -c 700 "$out/bench_216.json"; head -6 "$out/kernel_stats.txt"; cat "$out/traffic.json" | head -30
