#!/bin/bash
mkdir -p gpurun_out/s19
timeout 600 python tools/gpu_stream_latency.py --entries 1000 > gpurun_out/s19/latency.log 2>&1
cat gpurun_out/s19/latency.log
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/s19/tests.log
cat gpurun_out/s19/tests.log
