#!/bin/bash
mkdir -p gpurun_out/s34
timeout 600 python tools/gpu_stream_latency.py --entries 500 2>&1 | grep -v amdgpu | tee gpurun_out/s34/latency.log
timeout 900 python tools/gpu_scale.py 1 4 16 2>&1 | grep -v "amdgpu\|zlib\|gen \|oracle" | tee gpurun_out/s34/scale.log
timeout 900 python -m pytest tests/test_gpu_deflate.py -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/s34/tests.log
