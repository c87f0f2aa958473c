#!/bin/bash
mkdir -p gpurun_out/s17
timeout 900 python -m pytest tests/test_gpu_setlevel.py -x -q -m gpu 2>&1 | tail -30 > gpurun_out/s17/setlevel.log
cat gpurun_out/s17/setlevel.log
