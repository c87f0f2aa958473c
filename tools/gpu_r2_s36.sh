#!/bin/bash
mkdir -p gpurun_out/s36
timeout 900 python tools/gpu_matchlab.py --mib 256 SZL_SLICE=128 SZL_FTH2=24 SZL_FTH2=40 SZL_FTH2=48 SZL_VTH2=6 SZL_VTH2=12 SZL_QKEEP=40 SZL_QKEEP=56 SZL_QKEEP=64 SZL_VKEEP=2 SZL_VKEEP=8 SZL_FTH2=40,SZL_QKEEP=56 SZL_SLICE=192 > gpurun_out/s36/sweep.log 2>&1; grep -v amdgpu gpurun_out/s36/sweep.log
timeout 300 python tools/gpu_matchlab.py --mib 128 --kind logs --level 9 SZL_SLICE=512 SZL_SLICE=128 2>&1 | grep -v amdgpu | tee gpurun_out/s36/logs.log
