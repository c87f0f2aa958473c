#!/bin/bash
mkdir -p gpurun_out/s35
timeout 900 python tools/gpu_configs.py c4 2048 2>&1 | grep -v amdgpu | tee gpurun_out/s35/c4_full.log
SZL_INF_PAR_MAX_STREAMS=4096 timeout 900 python tools/gpu_configs.py c4 2048 2>&1 | grep -v amdgpu | sed 's/^/par: /' | tee -a gpurun_out/s35/c4_full.log
