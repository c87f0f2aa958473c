#!/bin/bash
mkdir -p gpurun_out/s41
timeout 300 python tools/gpu_matchlab.py --mib 64 --oracle SZL_MATCH_KERNEL=2 2>&1 | grep -v amdgpu | tee gpurun_out/s41/oracle.log
timeout 600 python tools/gpu_matchlab.py --mib 256 SZL_MATCH_KERNEL=2 SZL_FTH2=24 SZL_FTH2=28 2>&1 | grep -v amdgpu | tee gpurun_out/s41/t.log
