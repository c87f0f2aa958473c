#!/bin/bash
# round 2 session 15: two-byte QUICK filter in k_match4 and k_match5
mkdir -p gpurun_out/s15
timeout 200 python tools/gpu_matchlab.py --mib 64 --oracle SZL_MATCH_KERNEL=2 SZL_MATCH_KERNEL=3 > gpurun_out/s15/first.log 2>&1
echo "rc=$?" >> gpurun_out/s15/first.log
cat gpurun_out/s15/first.log
if grep -q "DIFFERS\|rc=124\|Error\|error" gpurun_out/s15/first.log; then exit 0; fi
timeout 300 python tools/gpu_matchlab.py --mib 128 --debug --reps 1 SZL_MATCH_KERNEL=2 SZL_MATCH_KERNEL=3 > gpurun_out/s15/counters.log 2>&1
grep "match4\|MATCH" gpurun_out/s15/counters.log
K=SZL_MATCH_KERNEL=3
J=SZL_MATCH_KERNEL=2
timeout 900 python tools/gpu_matchlab.py --mib 256 $J $K \
  $J,SZL_VTH2=8 $J,SZL_VTH2=16,SZL_VKEEP=8 $J,SZL_QKEEP=40 $J,SZL_QKEEP=56 $J,SZL_FTH2=24 $J,SZL_FTH2=40 $J,SZL_VKEEP=12 \
  $K,SZL_VKEEP5=16 $K,SZL_VKEEP5=32 $K,SZL_VTH5=24 $K,SZL_VTH5=56 $K,SZL_QKEEP5=48 $K,SZL_QKEEP5=80 $K,SZL_FTH5=24 $K,SZL_FTH5=40 $K,SZL_QMIN5=12 $K,SZL_QMIN5=40 \
  > gpurun_out/s15/sweep.log 2>&1
cat gpurun_out/s15/sweep.log
timeout 300 python tools/gpu_matchlab.py --mib 128 --kind logs --level 9 $J $K > gpurun_out/s15/logs9.log 2>&1
cat gpurun_out/s15/logs9.log
