#!/bin/bash
# GPU box: L2 / HBM traffic of the direct and Winograd fp32 trunk conv (tools/bin/convbench_f32)
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/wino_pmc2
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -s KILL 90 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $OUT/p1 -- tools/bin/convbench_f32 256 1 4 1 > $OUT/p1.log 2>&1 || exit 1
timeout -s KILL 90 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/p2 -- tools/bin/convbench_f32 256 1 4 1 > $OUT/p2.log 2>&1 || exit 1
timeout -s KILL 90 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $OUT/p3 -- tools/bin/convbench_f32 256 1 4 1 > $OUT/p3.log 2>&1 || echo p3 failed
python3 - <<'PY'
import csv, glob, collections, os
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/wino_pmc2'
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+'/**/*_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k='wino' if 'wino' in r['Kernel_Name'] else ('direct' if 'conv3x3' in r['Kernel_Name'] else 'other')
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in acc.items():
    print(k)
    for c,v in sorted(d.items()):
        print(f"  {c:32s} {sum(v)/len(v):16.0f}  n={len(v)}")
PY
