#!/bin/bash
mkdir -p gpurun_out/s23
SZL_C3_N=50000 timeout 600 python tools/gpu_configs.py c3 > gpurun_out/s23/c3.log 2>&1; grep -v amdgpu gpurun_out/s23/c3.log
SZL_LINKS=2 SZL_C3_N=50000 timeout 600 python tools/gpu_configs.py c3 2>&1 | grep "c3:" | tail -1 | sed 's/^/links2: /' | tee -a gpurun_out/s23/c3.log
timeout 600 python tools/gpu_configs.py c4 > gpurun_out/s23/c4.log 2>&1; grep -v amdgpu gpurun_out/s23/c4.log
