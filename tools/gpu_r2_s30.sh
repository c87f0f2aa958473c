#!/bin/bash
mkdir -p gpurun_out/s30
SZL_DEBUG=0 timeout 1200 python -m pytest tests/test_gpu_multi.py tests/test_gpu_window.py -x -q -m gpu 2>&1 | tail -30 > gpurun_out/s30/tests.log
cat gpurun_out/s30/tests.log
