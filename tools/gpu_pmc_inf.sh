#!/bin/bash
# SQ counters of k_inflate on one inflate call: tools/gpu_pmc_inf.sh <out-subdir> <n_members> <KiB each> [kind]
out=gpurun_out/$1; shift
mkdir -p $out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/$out -o p$i -- python $R/tools/gpu_lab.py inflate_one "$@" > $R/$out/run$i.txt 2> $R/$out/err$i.txt
done
cd $R
python3 - $out <<'PY'
import csv, glob, collections, json, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in sorted(glob.glob(out + '/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
res = {k: {c: int(v) for c, v in d.items()} for k, d in agg.items() if 'inflate' in k or 'k_find' in k or 'k_convert' in k or 'k_resolve' in k}
json.dump(res, open(out + '/pmc.json', 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
cat $out/run1.txt
