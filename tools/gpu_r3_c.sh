#!/bin/bash
mkdir -p gpurun_out/r3c
timeout 120 python tools/gpu_matchlab.py --mib 8 --oracle SZL_MATCH_KERNEL=5 2>&1 | grep -v "amdgpu" > gpurun_out/r3c/m5_8.log; cat gpurun_out/r3c/m5_8.log
timeout 200 python tools/gpu_matchlab.py --mib 128 --oracle SZL_MATCH_KERNEL=2 SZL_MATCH_KERNEL=5 SZL_MATCH_KERNEL=5,SZL_B5_LAB=1 SZL_MATCH_KERNEL=5,SZL_B5_LAB=3 SZL_MATCH_KERNEL=5,SZL_B5_LAB=7 SZL_MATCH_KERNEL=5,SZL_B5_LAB=31 2>&1 | grep -v amdgpu > gpurun_out/r3c/m5_lab.log; cat gpurun_out/r3c/m5_lab.log
timeout 200 python tools/gpu_matchlab.py --mib 64 --kind logs --level 9 --oracle SZL_MATCH_KERNEL=5 2>&1 | grep -v amdgpu > gpurun_out/r3c/m5_logs9.log; cat gpurun_out/r3c/m5_logs9.log
