#!/bin/bash
# SQ counters of the chunk-parallel Inflater on one 1 GiB text member (each rocprofv3 pass its own run, PMC never together with other trace
# domains):   bash tools/gpu_profile_inflate.sh   ->   gpurun_out/profile/pmc_sq_inflate_1gib.json (+ inflate_kernel_stats.csv)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profile
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/tools/gpu_inflate_big.py 1024"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/istats -o k -- $B > /dev/null 2> $O/istats.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/ipmc -o sq1 -- $B > /dev/null 2> $O/ipmc_sq1.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $O/ipmc -o sq2 -- $B > /dev/null 2> $O/ipmc_sq2.err
cd $R
python3 - <<'PY'
import csv, glob, collections, json, shutil
O = 'gpurun_out/profile'
f = (glob.glob(O + '/istats/**/*kernel_stats.csv', recursive=True) or [None])[0]
if f:
    shutil.copy(f, O + '/inflate_kernel_stats.csv')
    print(open(f).read()[:1500])
sq = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.Counter()
for tag in ('sq1', 'sq2'):
    for f in glob.glob(O + '/ipmc/**/%s_counter_collection.csv' % tag, recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0].replace('void ', '')
            sq[k][r['Counter_Name']] += float(r['Counter_Value'])
            if tag == 'sq1' and r['Counter_Name'] == 'SQ_WAVE_CYCLES': disp[k] += 1
out = {k: dict({c: int(v) for c, v in d.items()}, dispatches=disp.get(k, 0)) for k, d in sq.items() if k.startswith('szl::')}
json.dump(out, open(O + '/pmc_sq_inflate_1gib.json', 'w'), indent=1)
for k, d in out.items():
    if 'k_inflate' in k or 'k_find' in k: print(k, d)
PY
rm -rf $O/istats $O/ipmc
