#!/bin/bash
# full GPU suite + fuzz detail + bench line
mkdir -p gpurun_out/s7
export PYTHONUNBUFFERED=1
( time timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_inflate_fuzz.py --durations=8 ) > gpurun_out/s7/tests.log 2>&1
tail -15 gpurun_out/s7/tests.log
( timeout 600 python -m pytest tests/test_gpu_inflate_fuzz.py -m gpu -q ) > gpurun_out/s7/fuzz.log 2>&1
tail -60 gpurun_out/s7/fuzz.log
timeout 600 python bench.py --steps 5 --warmup 1 > gpurun_out/s7/bench.json 2> gpurun_out/s7/bench.err
cat gpurun_out/s7/bench.json; tail -3 gpurun_out/s7/bench.err
