#!/bin/bash
mkdir -p gpurun_out/r5
{
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/ktx
rocprofv3 --kernel-trace --output-format csv -d /tmp/ktx -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-dist-cg --no-slab-sweep --no-graph --steps 12 --phase-events none > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob
rows=[]
for f in glob.glob('/tmp/ktx/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
for f in glob.glob('/tmp/ktx/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "MEMCPY "+r.get("Direction","")))
rows.sort()
# find last 4 steps: locate k_patch_hex8_record occurrences
idx=[i for i,r in enumerate(rows) if 'k_patch_hex8_record' in r[2]]
start=idx[-5]
prev_end=None
for s,e,n in rows[start:idx[-1]+6]:
    gap = (s-prev_end)/1e3 if prev_end else 0.0
    print("gap %8.1f us  dur %9.1f us  %s" % (gap,(e-s)/1e3,n))
    prev_end=e
PY
} > gpurun_out/r5/u.log 2>&1
cat gpurun_out/r5/u.log | grep -v amdgpu.ids | tail -40
