#!/bin/bash
# HBM traffic of the dominant kernel: FETCH_SIZE and WRITE_SIZE in separate --pmc passes (TCC slots), kernel-trace only.
mkdir -p gpurun_out/traffic
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/traffic -o fetch -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/traffic/err1.txt
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/traffic -o write -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/traffic/err2.txt
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for tag in ('fetch', 'write'):
    f = glob.glob(f'gpurun_out/traffic/{tag}_counter_collection.csv')[0]
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        agg[k][0] += float(r['Counter_Value']); agg[k][1] += 1
    for k, (v, c) in agg.items():
        out.setdefault(k, {})[tag] = v
        out[k]['dispatches'] = c
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get('fetch', 0))[:8]:
    print(k, v)
json.dump(out, open('gpurun_out/traffic/traffic.json', 'w'), indent=1)
PY
