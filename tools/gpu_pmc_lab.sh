#!/bin/bash
# PMC passes (own runs, kernel-trace only) over the stage-B laboratory: tools/gpu_pmc_lab.sh <out-subdir> <mib> <cfg>
out=gpurun_out/$1; mib=$2; cfg=$3
mkdir -p $out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAVES TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/$out -o p$i -- python $R/tools/gpu_matchlab.py --mib $mib --reps 1 $cfg > /dev/null 2> $R/$out/err$i.txt
done
cd $R
python3 - $out <<'PY'
import csv, glob, collections, json, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in sorted(glob.glob(out + '/**/*counter_collection.csv', recursive=True)):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
res = {k: {c: int(v) for c, v in d.items()} for k, d in agg.items() if k.startswith('szl::k_match') or k.startswith('void szl::k_match')}
json.dump(res, open(out + '/pmc.json', 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
