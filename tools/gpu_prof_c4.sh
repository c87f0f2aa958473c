#!/bin/bash
mkdir -p gpurun_out/profc4
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/profc4 -o c4 -- python $GRAFT_REPO_ROOT/tools/gpu_configs.py c4 > $GRAFT_REPO_ROOT/gpurun_out/profc4/out.txt 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/profc4/c4_kernel_stats.csv')):
    ms=float(r['AverageNs'])/1e6
    if ms>0.3: print(r['Name'].split('(')[0][:40], r['Calls'], round(ms,2),'ms avg')
PY
