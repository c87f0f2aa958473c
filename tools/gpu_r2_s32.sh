#!/bin/bash
mkdir -p gpurun_out/s32
timeout 900 python tools/gpu_scale.py 1 4 16 64 256 2>&1 | grep -v "amdgpu\|zlib\|gen \|oracle" > gpurun_out/s32/scale.log; cat gpurun_out/s32/scale.log
timeout 600 python tools/gpu_stream_latency.py --entries 500 2>&1 | grep -v amdgpu | tee gpurun_out/s32/latency.log
SZL_C3_N=20000 timeout 600 python tools/gpu_configs.py c3 2>&1 | grep "c3:" | tee gpurun_out/s32/c3.log
timeout 900 python -m pytest tests/test_gpu_deflate.py tests/test_gpu_setlevel.py tests/test_gpu_window.py -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/s32/tests.log
