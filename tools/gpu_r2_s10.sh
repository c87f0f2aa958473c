#!/bin/bash
# round 2 session 10: parallel single-member inflate (first run), windowed path with pinned read-backs
mkdir -p gpurun_out/s10
timeout 600 python -m pytest tests/test_gpu_inflate_par.py -x -q -m gpu -s 2>&1 | tail -40 > gpurun_out/s10/par.log
timeout 300 python -m pytest tests/test_gpu_window.py tests/test_gpu_inflate.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/s10/window.log
SZL_WINDOW_FROM_KIB=0 timeout 300 python bench.py --steps 3 --warmup 1 > gpurun_out/s10/bench_windowed.json 2> gpurun_out/s10/bench_windowed.err
cat gpurun_out/s10/par.log gpurun_out/s10/window.log
tail -c 1500 gpurun_out/s10/bench_windowed.json
