#!/bin/bash
mkdir -p gpurun_out/s24
timeout 900 python -m pytest tests/test_gpu_inflate_par.py tests/test_gpu_inflate.py tests/test_gpu_inflate_fuzz.py tests/test_gpu_gzip.py tests/test_zipbatch.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/s24/tests.log
cat gpurun_out/s24/tests.log
SZL_DEBUG=1 timeout 600 python tools/gpu_configs.py c4 > gpurun_out/s24/c4.log 2>&1; grep -v "amdgpu\|match" gpurun_out/s24/c4.log | tail -12
timeout 600 python tools/gpu_inflate_perf.py 256 64 > gpurun_out/s24/inf.log 2>&1; grep -v amdgpu gpurun_out/s24/inf.log
