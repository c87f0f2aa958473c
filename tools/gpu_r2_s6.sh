#!/bin/bash
mkdir -p gpurun_out/s6
export PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
timeout 600 python tools/gpu_matchlab.py --mib 256 --oracle \
  SZL_MATCH_KERNEL=2,SZL_FTH2=32,SZL_VTH2=24,SZL_QKEEP=48,SZL_VKEEP=12 \
  SZL_MATCH_KERNEL=2,SZL_FTH2=32,SZL_VTH2=16,SZL_QKEEP=48,SZL_VKEEP=8 \
  SZL_MATCH_KERNEL=2,SZL_FTH2=32,SZL_VTH2=12,SZL_QKEEP=48,SZL_VKEEP=4 \
  SZL_MATCH_KERNEL=2,SZL_FTH2=32,SZL_VTH2=8,SZL_QKEEP=48,SZL_VKEEP=1 \
  SZL_MATCH_KERNEL=2,SZL_FTH2=24,SZL_VTH2=16,SZL_QKEEP=56,SZL_VKEEP=8 \
  SZL_MATCH_KERNEL=2,SZL_FTH2=24,SZL_VTH2=16,SZL_QKEEP=40,SZL_VKEEP=8 \
  SZL_MATCH_KERNEL=2,SZL_FTH2=20,SZL_VTH2=16,SZL_QKEEP=64,SZL_VKEEP=8 \
  SZL_MATCH_KERNEL=2,SZL_FTH2=40,SZL_VTH2=16,SZL_QKEEP=48,SZL_VKEEP=8 \
  SZL_MATCH_KERNEL=2,SZL_FTH2=28,SZL_VTH2=20,SZL_QKEEP=72,SZL_VKEEP=8 \
  > gpurun_out/s6/lab_sweep.log 2>&1
cat gpurun_out/s6/lab_sweep.log
cd /tmp && export TMPDIR=/tmp
for cfg in SZL_MATCH_KERNEL=2; do
  tag=$(echo $cfg | tr ',=' '__')
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL --kernel-trace --output-format csv -d $R/gpurun_out/s6 -o a_$tag -- python $R/tools/gpu_matchlab.py --mib 128 --reps 1 $cfg > /dev/null 2> $R/gpurun_out/s6/err_a_$tag.txt
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $R/gpurun_out/s6 -o b_$tag -- python $R/tools/gpu_matchlab.py --mib 128 --reps 1 $cfg > /dev/null 2> $R/gpurun_out/s6/err_b_$tag.txt
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/s6/*counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    for k in agg:
        if 'k_match' in k and 'lazy' not in k:
            print('  ', k[:40], {c: int(v) for c, v in sorted(agg[k].items())})
PY
