#!/bin/bash
mkdir -p gpurun_out/pmcinf
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcinf -o p1 -- python $GRAFT_REPO_ROOT/tools/gpu_configs.py c4 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcinf -o p2 -- python $GRAFT_REPO_ROOT/tools/gpu_configs.py c4 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmcinf/*counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    for k in agg:
        if 'k_inflate' in k: print(k, {c:int(v) for c,v in agg[k].items()})
PY
