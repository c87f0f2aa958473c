#!/bin/bash
mkdir -p gpurun_out/s26
timeout 900 python -m pytest tests/test_gpu_inflate_par.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/s26/tests.log
cat gpurun_out/s26/tests.log
SZL_DEBUG=1 timeout 600 python tools/gpu_configs.py c4 > gpurun_out/s26/c4.log 2>&1; grep -v "amdgpu\|match\|stage B" gpurun_out/s26/c4.log | tail -9
