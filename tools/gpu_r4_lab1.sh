#!/bin/bash
# round 4, stage-B A/B: k_match4 (SZL_B9=0) vs the all-assembly k_match9 (SZL_B9=1) on the shipped library
mkdir -p gpurun_out; cd "$(dirname "$0")/.."
timeout 420 python tools/gpu_matchlab.py --product --mib 256 --oracle SZL_B9=0 SZL_B9=1 "SZL_B9=1,SZL9_FTH=16" "SZL_B9=1,SZL9_FTH=32" "SZL_B9=1,SZL9_QKEEP=48" "SZL_B9=1,SZL9_QKEEP=80" "SZL_B9=1,SZL9_KTAIL=1" "SZL_B9=1,SZL9_KTAIL=4" "SZL_B9=1,SZL_SLICE=64" "SZL_B9=1,SZL_SLICE=256" > gpurun_out/lab1.log 2>&1
echo "lab1 rc $?" >> gpurun_out/lab1.log
timeout 200 python tools/gpu_matchlab.py --product --mib 256 --debug --reps 1 SZL_B9=0 SZL_B9=1 > gpurun_out/lab1_dbg.log 2>&1
echo "dbg rc $?" >> gpurun_out/lab1_dbg.log
timeout 200 python tools/gpu_matchlab.py --product --mib 256 --kind logs --level 9 --reps 2 SZL_B9=0 SZL_B9=1 > gpurun_out/lab1_logs9.log 2>&1
SZL_B9=1 timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_b9.log 2>&1
echo "pytest rc $?" >> gpurun_out/pytest_b9.log
tail -n 30 gpurun_out/lab1.log gpurun_out/lab1_dbg.log gpurun_out/lab1_logs9.log; tail -n 8 gpurun_out/pytest_b9.log
