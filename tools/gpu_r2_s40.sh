#!/bin/bash
# final evidence: kernel stats of the bench command, HBM traffic passes, SQ counter passes (final code)
mkdir -p gpurun_out/s40
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/s40/stats -o r02 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/s40/bench_prof.json 2> $R/gpurun_out/s40/rocprof.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/s40/traffic -o fetch -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/s40/err_fetch.txt
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/s40/traffic -o write -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/s40/err_write.txt
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/s40/pmc -o p1 -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/s40/err_p1.txt
timeout 600 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $R/gpurun_out/s40/pmc -o p2 -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/s40/err_p2.txt
cd $R
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for tag in ('fetch', 'write'):
    fs = glob.glob(f'gpurun_out/s40/traffic/**/{tag}_counter_collection.csv', recursive=True)
    if not fs: continue
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        agg[k][0] += float(r['Counter_Value']); agg[k][1] += 1
    for k, (v, c) in agg.items():
        out.setdefault(k, {})[tag] = v
        out[k]['dispatches'] = c
json.dump(out, open('gpurun_out/s40/traffic_pmc.json', 'w'), indent=1)
pm = {}
for f in sorted(glob.glob('gpurun_out/s40/pmc/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        pm.setdefault(k, {}).setdefault(r['Counter_Name'], 0.0)
        pm[k][r['Counter_Name']] += float(r['Counter_Value'])
keep = {k: {c: int(v) for c, v in d.items()} for k, d in pm.items() if any(t in k for t in ('k_match4', 'k_links3', 'k_spec_win', 'k_emit_copy', 'k_block_encode'))}
json.dump(keep, open('gpurun_out/s40/pmc_sq.json', 'w'), indent=1)
print(json.dumps(keep.get('szl::k_match4<false>', {}), indent=0))
print({k: v for k, v in out.items() if 'k_match4' in k})
PY
f=$(find gpurun_out/s40/stats -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/s40/kernel_stats.csv; head -6 "$f" | cut -c1-120
tail -c 900 gpurun_out/s40/bench_prof.json
find gpurun_out/s40 -name "*trace.csv" -size +8M -delete
find gpurun_out/s40 -name "*counter_collection.csv" -size +8M -delete
