#!/bin/bash
# GPU box: bench + rocprofv3 kernel trace + PMC passes of the discriminator step (tools/bench_disc.py).
# Output -> gpurun_out/prof_disc_<tag>/ ; copy the summaries you want kept into profiles/.
# usage: bash tools/prof_disc.sh <tag>
set -e
TAG=${1:-r01}
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_disc_$TAG
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 python3 tools/bench_disc.py > $OUT/bench.json
cat $OUT/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/bench_disc.py --steps 5 --warmup 2 > $OUT/trace.log 2>&1
python3 tools/disc_layers.py $OUT/trace | tee $OUT/layers.txt
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_fetch -- python3 tools/bench_disc.py --steps 1 --warmup 1 > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- python3 tools/bench_disc.py --steps 1 --warmup 1 > $OUT/pmc_write.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/pmc_sq -- python3 tools/bench_disc.py --steps 1 --warmup 1 > $OUT/pmc_sq.log 2>&1 || echo "sq pass failed"
echo done
