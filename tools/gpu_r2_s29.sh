#!/bin/bash
mkdir -p gpurun_out/s29
timeout 600 python tools/gpu_hostpath.py > gpurun_out/s29/hostpath.log 2>&1; grep -v amdgpu gpurun_out/s29/hostpath.log
