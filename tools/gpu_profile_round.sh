#!/bin/bash
# Everything profiles/rNN/ holds for the headline command, in one GPU call (each rocprofv3 pass is its own run: kernel trace + stats,
# FETCH_SIZE, WRITE_SIZE, two SQ counter sets — never PMC together with other trace domains):
#   bash tools/gpu_profile_round.sh          ->  gpurun_out/profile/{bench.json,bench_kernel_stats.csv,traffic_pmc.json,pmc_sq_1gib.json}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profile
mkdir -p $O
python $R/bench.py > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-extra-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o k -- $B --steps 3 --warmup 1 > $O/bench_under_rocprof.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc -o fetch -- $B --steps 1 --warmup 0 > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc -o write -- $B --steps 1 --warmup 0 > /dev/null 2> $O/pmc_write.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/pmc -o sq1 -- $B --steps 1 --warmup 0 > /dev/null 2> $O/pmc_sq1.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $O/pmc -o sq2 -- $B --steps 1 --warmup 0 > /dev/null 2> $O/pmc_sq2.err
cd $R
python3 - <<'PY'
import csv, glob, collections, json, shutil
O = 'gpurun_out/profile'
f = (glob.glob(O + '/stats/**/*kernel_stats.csv', recursive=True) or [None])[0]
if f:
    shutil.copy(f, O + '/bench_kernel_stats.csv')
    print(open(f).read()[:1800])
tr = {}
for tag in ('fetch', 'write'):
    for f in glob.glob(O + '/pmc/**/%s_counter_collection.csv' % tag, recursive=True):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0].replace('void ', '')
            agg[k][0] += float(r['Counter_Value']); agg[k][1] += 1
        for k, (v, c) in agg.items():
            tr.setdefault(k, {})[tag] = v
            tr[k]['dispatches'] = c
json.dump(tr, open(O + '/traffic_pmc.json', 'w'), indent=1)
sq = collections.defaultdict(lambda: collections.defaultdict(float))
for tag in ('sq1', 'sq2'):
    for f in glob.glob(O + '/pmc/**/%s_counter_collection.csv' % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            sq[r['Kernel_Name'].split('(')[0].replace('void ', '')][r['Counter_Name']] += float(r['Counter_Value'])
json.dump({k: {c: int(v) for c, v in d.items()} for k, d in sq.items() if k.startswith('szl::')}, open(O + '/pmc_sq_1gib.json', 'w'), indent=1)
for k in ('szl::k_match9<false>', 'szl::k_spec_win<32>', 'szl::k_links3'):
    print(k, tr.get(k), dict(sq.get(k, {})))
PY
rm -rf $O/stats $O/pmc
# the other measured paths of the round (logs next to the headline's)
python $R/tools/gpu_inflate_ab.py libszl_amd.so 2>&1 | grep -v "^\[szl\]" > $O/inflate_round.log
python $R/tools/gpu_inflate_big.py 1024 2>&1 | grep -v "stage B\|links\|match_ms" >> $O/inflate_round.log
python $R/tools/gpu_inflate_big.py 1024 logs 2>&1 | tail -1 >> $O/inflate_round.log
python $R/tools/gpu_small_call.py 200 > $O/small_calls.log 2>&1
timeout 300 python $R/tools/gpu_lab.py read_path --mib 512 > $O/read_path.log 2>&1     # the drop-in read path by constructor (round 5)
timeout 300 python $R/tools/gpu_lab.py write_path > $O/write_path.log 2>&1             # GZipOutputStream by write size
timeout 300 python $R/tools/gpu_stream_latency.py > $O/stream_latency.log 2>&1
if [ -n "${SZL_PROFILE_LEVELS_1_4:-}" ]; then python $R/tools/gpu_fast.py 4 4000 > $O/levels_1_4.log 2>&1; fi   # (unchanged since round 3: on request)
tail -n 30 $O/inflate_round.log $O/small_calls.log $O/read_path.log $O/write_path.log
