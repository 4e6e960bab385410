#!/bin/bash
# GPU box: PMC passes over tools/bin/convbench (v1 and v2 trunk conv kernels)
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/conv_pmc
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $OUT/p1 -- tools/bin/convbench 256 1 4 1 > $OUT/p1.log 2>&1 || exit 1
timeout -s KILL 90 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/p2 -- tools/bin/convbench 256 1 4 1 > $OUT/p2.log 2>&1 || exit 1
timeout -s KILL 90 rocprofv3 --pmc SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VALU --output-format csv -d $OUT/p3 -- tools/bin/convbench 256 1 4 1 > $OUT/p3.log 2>&1 || echo "p3 failed"
echo done
python3 - <<'PY'
import csv, glob, collections, os
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/conv_pmc'
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+'/**/*_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k='v2' if 'c319conv3x3' in r['Kernel_Name'] or 'c3::conv3x3' in r['Kernel_Name'] else ('v1' if 'conv3x3' in r['Kernel_Name'] else 'other')
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in acc.items():
    print(k)
    for c,v in sorted(d.items()):
        print(f"  {c:32s} {sum(v)/len(v):16.0f}  n={len(v)}")
PY
