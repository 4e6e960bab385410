#!/bin/bash
# GPU box: rocprofv3 kernel trace of the GAN iteration (bench.py --workload cfg4), all three arithmetics (PRECS="bf16" for one).
# usage: bash tools/prof_cfg4.sh <tag>   -> gpurun_out/prof_cfg4_<tag>/{fp32,f16,bf16}_kernel_stats.csv + bench lines
set -e
TAG=${1:-r03}
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_cfg4_$TAG
rm -rf $OUT; mkdir -p $OUT
for P in ${PRECS:-fp32 f16 bf16}; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${P}_trace -- python3 bench.py --workload cfg4 --train-precision $P --steps 10 --warmup 3 --no-cpu-baseline > $OUT/${P}_trace.log 2>&1
  grep -h '^{"metric"' $OUT/${P}_trace.log > $OUT/${P}_bench_under_rocprof.json || true
  f=$(find $OUT/${P}_trace -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $OUT/${P}_kernel_stats.csv
done
echo done
