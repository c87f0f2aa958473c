#!/bin/bash
mkdir -p gpurun_out/s37
K=SZL_VTH2=2,SZL_QKEEP=64,SZL_VKEEP=2
timeout 900 python tools/gpu_matchlab.py --mib 256 $K SZL_VTH2=2,SZL_QKEEP=64,SZL_VKEEP=1 SZL_VTH2=3,SZL_QKEEP=64,SZL_VKEEP=2 SZL_VTH2=2,SZL_QKEEP=56,SZL_VKEEP=2 SZL_VTH2=2,SZL_QKEEP=72,SZL_VKEEP=2 $K,SZL_FTH2=28 $K,SZL_FTH2=36 $K,SZL_FTH2=20 SZL_VTH2=4,SZL_QKEEP=60,SZL_VKEEP=1 > gpurun_out/s37/sweep3.log 2>&1; grep -v amdgpu gpurun_out/s37/sweep3.log
timeout 300 python tools/gpu_matchlab.py --mib 128 --kind logs --level 9 $K SZL_VTH2=8,SZL_QKEEP=48,SZL_VKEEP=4 2>&1 | grep -v amdgpu | tee gpurun_out/s37/logs3.log
timeout 300 python tools/gpu_matchlab.py --mib 128 --kind dickens --level 8 $K SZL_VTH2=8,SZL_QKEEP=48,SZL_VKEEP=4 2>&1 | grep -v amdgpu | tee gpurun_out/s37/dickens3.log
