#!/bin/bash
# after the longer stage-B tiles: one stream by size, and the latency of one 64 KiB entry through the streaming object
mkdir -p gpurun_out/s52
timeout 110 python tools/gpu_scale.py 1 4 16 64 256 2>&1 | grep -v "amdgpu\|zlib\|gen \|oracle" > gpurun_out/s52/scale.log; cat gpurun_out/s52/scale.log
timeout 50 python tools/gpu_stream_latency.py --entries 300 2>&1 | grep -v amdgpu > gpurun_out/s52/latency.log; cat gpurun_out/s52/latency.log
