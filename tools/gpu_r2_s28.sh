#!/bin/bash
mkdir -p gpurun_out/s28
timeout 900 python -m pytest tests/test_gpu_inflate_par.py tests/test_gpu_inflate.py tests/test_gpu_inflate_fuzz.py tests/test_gpu_gzip.py tests/test_zipbatch.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/s28/tests.log
cat gpurun_out/s28/tests.log
SZL_DEBUG=1 timeout 600 python tools/gpu_inflate_perf.py 1024 2048 65536 > gpurun_out/s28/inf.log 2>&1; grep -v "amdgpu\|match\|stage B" gpurun_out/s28/inf.log | tail -22
SZL_C3_N=50000 timeout 600 python tools/gpu_configs.py c3 2>&1 | grep "inflate" | tee gpurun_out/s28/c3.log
timeout 600 python tools/gpu_configs.py c4 2>&1 | grep "inflate" | tee gpurun_out/s28/c4.log
