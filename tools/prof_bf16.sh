#!/bin/bash
# run on the GPU box: kernel trace + two PMC passes for the bf16 path
set -e
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bf16_trace -- python3 bench.py --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof_bf16_trace.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $OUT/prof_bf16_pmc1 -- python3 bench.py --dtype bf16 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/prof_bf16_pmc1.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VALU SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/prof_bf16_pmc2 -- python3 bench.py --dtype bf16 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/prof_bf16_pmc2.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/prof_bf16_pmc3 -- python3 bench.py --dtype bf16 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/prof_bf16_pmc3.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/prof_bf16_pmc4 -- python3 bench.py --dtype bf16 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/prof_bf16_pmc4.log 2>&1
echo done
