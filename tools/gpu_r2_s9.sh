#!/bin/bash
mkdir -p gpurun_out/s9
export PYTHONUNBUFFERED=1
timeout 600 python tools/gpu_hostpath.py > gpurun_out/s9/hostpath.log 2>&1; cat gpurun_out/s9/hostpath.log
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/s9/bench_windowed.json 2> gpurun_out/s9/bench.err; cat gpurun_out/s9/bench_windowed.json
SZL_WINDOW_KIB=4194304 timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/s9/bench_monolithic.json 2>> gpurun_out/s9/bench.err; cat gpurun_out/s9/bench_monolithic.json
