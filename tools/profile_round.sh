#!/bin/bash
# GPU box: rocprofv3 kernel trace + PMC passes for bench.py, both dtypes.  Output -> gpurun_out/prof_<tag>/
# usage: bash tools/profile_round.sh <tag>
set -e
TAG=${1:-r01}
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
for DT in f32 bf16; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${DT}_trace -- python3 bench.py --dtype $DT --steps 5 --warmup 2 --no-cpu-baseline --lean > $OUT/${DT}_trace.log 2>&1
  timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/${DT}_pmc_fetch -- python3 bench.py --dtype $DT --steps 1 --warmup 1 --no-cpu-baseline --lean > $OUT/${DT}_pmc_fetch.log 2>&1
  timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/${DT}_pmc_write -- python3 bench.py --dtype $DT --steps 1 --warmup 1 --no-cpu-baseline --lean > $OUT/${DT}_pmc_write.log 2>&1
  timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d $OUT/${DT}_pmc_sq -- python3 bench.py --dtype $DT --steps 1 --warmup 1 --no-cpu-baseline --lean > $OUT/${DT}_pmc_sq.log 2>&1 || echo "sq pass failed for $DT"
  grep -h '^{"metric"' $OUT/${DT}_trace.log > $OUT/${DT}_bench_under_rocprof.json || true
done
rocm-smi --showpower --showmaxpower --showclocks > $OUT/rocm_smi.txt 2>&1 || true
echo done
