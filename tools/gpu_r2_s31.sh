#!/bin/bash
mkdir -p gpurun_out/s31
SZL_DEBUG=1 timeout 900 python tools/gpu_multi_stream.py 512 2 6 2>&1 | grep -v "amdgpu\|match\|stage B" > gpurun_out/s31/multi.log
SZL_DEBUG=1 timeout 900 python tools/gpu_multi_stream.py 512 4 6 2>&1 | grep -v "amdgpu\|match\|stage B" >> gpurun_out/s31/multi.log
cat gpurun_out/s31/multi.log
