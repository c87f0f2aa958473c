#!/bin/bash
mkdir -p gpurun_out/s14
K=SZL_MATCH_KERNEL=3
timeout 900 python tools/gpu_matchlab.py --mib 256 SZL_MATCH_KERNEL=2 $K \
  $K,SZL_VKEEP5=12 $K,SZL_VKEEP5=24 $K,SZL_VKEEP5=40 $K,SZL_VKEEP5=64 \
  $K,SZL_VKEEP5=32,SZL_VTH5=24 $K,SZL_VKEEP5=32,SZL_VTH5=32 $K,SZL_VKEEP5=32,SZL_VTH5=48 \
  $K,SZL_VKEEP5=32,SZL_QKEEP5=48 $K,SZL_VKEEP5=32,SZL_QKEEP5=56 $K,SZL_VKEEP5=32,SZL_QKEEP5=72 \
  $K,SZL_VKEEP5=32,SZL_FTH5=24 $K,SZL_VKEEP5=32,SZL_FTH5=40 $K,SZL_VKEEP5=32,SZL_FTH5=16 \
  $K,SZL_VKEEP5=32,SZL_QMIN5=8 $K,SZL_VKEEP5=32,SZL_QMIN5=16 $K,SZL_VKEEP5=32,SZL_QMIN5=40 \
  > gpurun_out/s14/sweep.log 2>&1
cat gpurun_out/s14/sweep.log
timeout 300 python tools/gpu_matchlab.py --mib 128 --debug --reps 1 $K,SZL_VKEEP5=32 > gpurun_out/s14/counters.log 2>&1
grep "match4\|MATCH" gpurun_out/s14/counters.log
