#!/bin/bash
# end-of-round check as the driver runs it (GPU tests, smoke(), bench line), then the rocprofv3 summaries the bench line refers to
mkdir -p gpurun_out/final gpurun_out/prof gpurun_out/traffic
( time timeout 260 python -m pytest tests -x -q -m gpu ) 2>&1 | tail -8 > gpurun_out/final/tests.log
cat gpurun_out/final/tests.log
timeout 90 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 > gpurun_out/final/smoke.log; cat gpurun_out/final/smoke.log
timeout 150 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err; tail -c 2500 gpurun_out/final/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r02 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof/bench_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof/rocprof.err
find $GRAFT_REPO_ROOT/gpurun_out/prof -name "*kernel_trace.csv" -delete
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof -name "*kernel_stats.csv" | head -1); head -8 "$f" | sed 's/(.*)"/"/' | cut -c1-150
timeout 70 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/traffic -o fetch -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/traffic/err1.txt
timeout 70 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/traffic -o write -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/traffic/err2.txt
cd $GRAFT_REPO_ROOT
find gpurun_out/traffic -name "*kernel_trace.csv" -delete
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for tag in ('fetch', 'write'):
    fs = glob.glob(f'gpurun_out/traffic/**/{tag}_counter_collection.csv', recursive=True)
    if not fs: print("no", tag); continue
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        agg[k][0] += float(r['Counter_Value']); agg[k][1] += 1
    for k, (v, c) in agg.items():
        out.setdefault(k, {})[tag] = v
        out[k]['dispatches'] = c
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get('fetch', 0))[:6]:
    print(k, v)
json.dump(out, open('gpurun_out/traffic/traffic.json', 'w'), indent=1)
PY
find gpurun_out/traffic -name "*counter_collection.csv" -size +4M -delete
