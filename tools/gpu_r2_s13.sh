#!/bin/bash
# round 2 session 13: run-ahead stage-B engine (k_match5), first run + sweep
mkdir -p gpurun_out/s13
timeout 200 python tools/gpu_matchlab.py --mib 64 --oracle SZL_MATCH_KERNEL=2 SZL_MATCH_KERNEL=3 > gpurun_out/s13/first.log 2>&1
echo "rc=$?" >> gpurun_out/s13/first.log
cat gpurun_out/s13/first.log
if grep -q "DIFFERS\|rc=124\|Error\|error" gpurun_out/s13/first.log; then exit 0; fi
timeout 300 python tools/gpu_matchlab.py --mib 128 --debug --reps 1 SZL_MATCH_KERNEL=3 > gpurun_out/s13/counters.log 2>&1
grep "match4\|MATCH" gpurun_out/s13/counters.log
K=SZL_MATCH_KERNEL=3
timeout 900 python tools/gpu_matchlab.py --mib 256 SZL_MATCH_KERNEL=2 $K \
  $K,SZL_VTH5=24 $K,SZL_VTH5=56 $K,SZL_VTH5=72 \
  $K,SZL_QKEEP5=48 $K,SZL_QKEEP5=80 $K,SZL_QKEEP5=96 \
  $K,SZL_FTH5=24 $K,SZL_FTH5=48 \
  $K,SZL_QMIN5=8 $K,SZL_QMIN5=48 \
  $K,SZL_VKEEP5=1 $K,SZL_VKEEP5=12 \
  > gpurun_out/s13/sweep.log 2>&1
cat gpurun_out/s13/sweep.log
timeout 300 python tools/gpu_matchlab.py --mib 128 --kind logs --level 9 SZL_MATCH_KERNEL=2 $K > gpurun_out/s13/logs9.log 2>&1
cat gpurun_out/s13/logs9.log
