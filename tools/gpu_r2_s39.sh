#!/bin/bash
mkdir -p gpurun_out/s39
timeout 300 python tools/gpu_matchlab.py --mib 64 --oracle SZL_MATCH_KERNEL=2 2>&1 | grep -v amdgpu | tee gpurun_out/s39/oracle.log
timeout 900 python tools/gpu_matchlab.py --mib 256 SZL_MATCH_KERNEL=2 SZL_VTH2=1 SZL_VTH2=3 SZL_VKEEP=1 SZL_VKEEP=3 SZL_QKEEP=56 SZL_QKEEP=72 SZL_FTH2=24 SZL_FTH2=40 2>&1 | grep -v amdgpu | tee gpurun_out/s39/sweep.log
timeout 300 python tools/gpu_matchlab.py --mib 128 --kind logs --level 9 SZL_MATCH_KERNEL=2 2>&1 | grep -v amdgpu | tee gpurun_out/s39/logs.log
