#!/bin/bash
# what bounds stage B?  stage-only / short-chain timing experiments (outputs are NOT valid deflate here)
mkdir -p gpurun_out/s4
export PYTHONUNBUFFERED=1
timeout 600 python tools/gpu_matchlab.py --mib 256 --debug --reps 2 SZL_MATCH_KERNEL=2,SZL_B_EXP=1 SZL_MATCH_KERNEL=2,SZL_B_EXP=0,SZL_B_CHAIN=1 SZL_MATCH_KERNEL=2,SZL_B_CHAIN=4 SZL_MATCH_KERNEL=2,SZL_B_CHAIN=16 SZL_MATCH_KERNEL=2,SZL_B_CHAIN=32 SZL_MATCH_KERNEL=2,SZL_B_CHAIN=64 SZL_MATCH_KERNEL=2,SZL_B_CHAIN=128 2>&1 | grep -v "^\[szl\] match:\|stage B" > gpurun_out/s4/lab_exp.log
cat gpurun_out/s4/lab_exp.log
