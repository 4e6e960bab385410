#!/bin/bash
# GPU box: PMC passes for the pre-train step (tools/bench_pretrain.py), both conv arithmetics.  Output -> gpurun_out/prof_train_<tag>/
# usage: bash tools/prof_train.sh <tag>
set -e
TAG=${1:-r02w}
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_train_$TAG
rm -rf $OUT; mkdir -p $OUT
for P in fp32 f16; do
  timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS --output-format csv -d $OUT/${P}_sq -- python3 tools/bench_pretrain.py --precision $P --steps 2 --warmup 1 > $OUT/${P}_sq.log 2>&1
  timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/${P}_fetch -- python3 tools/bench_pretrain.py --precision $P --steps 2 --warmup 1 > $OUT/${P}_fetch.log 2>&1
  timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/${P}_write -- python3 tools/bench_pretrain.py --precision $P --steps 2 --warmup 1 > $OUT/${P}_write.log 2>&1
done
echo done
