#!/bin/bash
# GPU box: time and L2<->fabric traffic of the trunk convs in one go (lean cfg1 legs).  Usage: bash tools/prof_traffic_quick.sh TAG
set -e
TAG=${1:-quick}
R=$PWD
OUT=$R/gpurun_out/traffic_$TAG
mkdir -p $OUT
for D in f32 bf16; do
  timeout -k 10 120 python3 bench.py --lean --dtype $D --steps 20 --warmup 5 > $OUT/bench_$D.json 2> $OUT/bench_$D.err
  tail -c 400 $OUT/bench_$D.json; echo
done
cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
for D in f32 bf16; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${D}_$C -o p -- python3 $R/bench.py --lean --dtype $D --steps 2 --warmup 1 > $OUT/${D}_$C.log 2>&1
  done
done
python3 - <<PY
import csv, collections
for D in ("f32", "bf16"):
    for C in ("FETCH_SIZE", "WRITE_SIZE"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open("$OUT/%s_%s/p_counter_collection.csv" % (D, C))):
            if "conv3x3" in r["Kernel_Name"]:
                agg[r["Kernel_Name"].split("(")[0][-60:]].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            print(D, C, k, len(v), round(sum(v) / len(v) * (2 if C == "FETCH_SIZE" else 1) * 1024 / 1e6, 1), "MB per launch")
PY
