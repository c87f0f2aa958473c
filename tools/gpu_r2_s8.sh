#!/bin/bash
mkdir -p gpurun_out/s8
export PYTHONUNBUFFERED=1
( time timeout 1200 python -m pytest tests/test_gpu_window.py tests/test_gpu_multi.py tests/test_zipbatch.py tests/test_crypto_hook.py tests/test_gpu_inflate_fuzz.py -m gpu -q --durations=6 ) > gpurun_out/s8/tests.log 2>&1
tail -40 gpurun_out/s8/tests.log
