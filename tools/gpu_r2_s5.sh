#!/bin/bash
# PMC counters of k_match4 at chain budgets 1 and 128 (what is the fixed per-position cost made of?)
mkdir -p gpurun_out/s5
export PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cfg in SZL_MATCH_KERNEL=2,SZL_B_CHAIN=1 SZL_MATCH_KERNEL=2 SZL_MATCH_KERNEL=1; do
  tag=$(echo $cfg | tr ',=' '__')
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL --kernel-trace --output-format csv -d $R/gpurun_out/s5 -o a_$tag -- python $R/tools/gpu_matchlab.py --mib 128 --reps 1 $cfg > /dev/null 2> $R/gpurun_out/s5/err_a_$tag.txt
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $R/gpurun_out/s5 -o b_$tag -- python $R/tools/gpu_matchlab.py --mib 128 --reps 1 $cfg > /dev/null 2> $R/gpurun_out/s5/err_b_$tag.txt
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/s5/*counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    print(f)
    for k in agg:
        if 'k_match' in k and 'lazy' not in k:
            print('  ', k[:40], {c: int(v) for c, v in sorted(agg[k].items())})
PY
tail -3 gpurun_out/s5/err_a_SZL_MATCH_KERNEL_2.txt
