#!/bin/bash
mkdir -p gpurun_out/r5
{
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/ktx
rocprofv3 --kernel-trace --output-format csv -d /tmp/ktx -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-dist-cg --no-slab-sweep --steps 12 --warmup 3 --phase-events none > /tmp/b.json 2>/dev/null
tail -1 /tmp/b.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('plain ms/step %.3f graph %.3f' % (d['ms_per_step'], d['graph_step']['ms_per_step']))"
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob
rows=[]
for f in glob.glob('/tmp/ktx/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]))
rows.sort()
idx=[i for i,r in enumerate(rows) if 'k_patch_hex8_record' in r[2]]
print(len(idx), "M+K launches")
for j,i in enumerate(idx):
    nxt = idx[j+1] if j+1 < len(idx) else None
    names = {}
    seg = rows[i:(nxt if nxt else i+6)]
    per = (rows[nxt][0]-rows[i][0])/1e3 if nxt else float('nan')
    ds = {r[2].split('(')[0].replace('void tb::','')[:22]: round((r[1]-r[0])/1e3,1) for r in seg}
    print("launch %2d  period %9.1f us  sum-of-durations %8.1f  %s" % (j, per, sum((r[1]-r[0]) for r in seg)/1e3, ds))
PY
} > gpurun_out/r5/x.log 2>&1
grep -v amdgpu.ids gpurun_out/r5/x.log | tail -45
