#!/bin/bash
mkdir -p gpurun_out/s22
timeout 600 python -m pytest tests/test_gpu_gzip.py -x -q -m gpu 2>&1 | tail -30 > gpurun_out/s22/gzip.log
cat gpurun_out/s22/gzip.log
