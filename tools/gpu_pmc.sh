#!/bin/bash
# PMC pass (own run, kernel-trace only) over a reduced workload
mkdir -p gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc -o p1 -- python $GRAFT_REPO_ROOT/bench.py --mib 256 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/pmc/err1.txt
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc -o p2 -- python $GRAFT_REPO_ROOT/bench.py --mib 256 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/pmc/err2.txt
cd $GRAFT_REPO_ROOT
ls gpurun_out/pmc
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc/*counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    print(f)
    for k in ('szl::k_match','szl::k_links','szl::k_emit','szl::k_spec'):
        if k in agg: print(' ', k, {c:int(v) for c,v in agg[k].items()})
PY
tail -3 gpurun_out/pmc/err1.txt
