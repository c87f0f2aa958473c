#!/bin/bash
mkdir -p gpurun_out/s25
SZL_DEBUG=1 timeout 600 python tools/gpu_configs.py c4 > gpurun_out/s25/c4.log 2>&1; grep -v amdgpu gpurun_out/s25/c4.log | tail -30
