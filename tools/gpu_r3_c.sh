#!/bin/bash
mkdir -p gpurun_out/r3c
timeout 120 python tools/gpu_matchlab.py --mib 8 --oracle SZL_MATCH_KERNEL=5 2>&1 | grep -v "amdgpu" > gpurun_out/r3c/m5_8.log; cat gpurun_out/r3c/m5_8.log
timeout 200 python tools/gpu_matchlab.py --mib 64 --reps 1 --oracle --debug SZL_MATCH_KERNEL=5 2>&1 | grep -v "amdgpu\|match:\|stage B" > gpurun_out/r3c/m5_phases.log; cat gpurun_out/r3c/m5_phases.log
timeout 200 python tools/gpu_matchlab.py --mib 64 --kind logs --level 9 --oracle SZL_MATCH_KERNEL=5 2>&1 | grep -v amdgpu > gpurun_out/r3c/m5_logs9.log; cat gpurun_out/r3c/m5_logs9.log
timeout 200 python tools/gpu_matchlab.py --mib 64 --kind logs --level 6 --oracle SZL_MATCH_KERNEL=2 SZL_MATCH_KERNEL=5 2>&1 | grep -v amdgpu > gpurun_out/r3c/m5_logs6.log; cat gpurun_out/r3c/m5_logs6.log
timeout 200 python tools/gpu_matchlab.py --mib 256 SZL_MATCH_KERNEL=2 SZL_MATCH_KERNEL=5 2>&1 | grep -v amdgpu > gpurun_out/r3c/m5_256.log; cat gpurun_out/r3c/m5_256.log
