#!/bin/bash
# round 4, GPU call 5: the exact decoder (k_inflate_exact) against the fuzz corpora, the streaming Inflater's long-input path, InflaterInputStream by buffer size
mkdir -p gpurun_out; cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_inflate_fuzz.py tests/test_gpu_inflate_stream_bulk.py tests/test_gpu_inflate.py tests/test_gpu_inflate_par.py tests/test_gpu_gzip.py -m gpu -q > gpurun_out/c5_inflate.log 2>&1
echo "inflate rc $?" >> gpurun_out/c5_inflate.log
timeout 400 python tools/gpu_stream_latency.py --entries 300 > gpurun_out/c5_stream_latency.log 2>&1
echo "latency rc $?" >> gpurun_out/c5_stream_latency.log
tail -n 30 gpurun_out/c5_inflate.log; tail -n 12 gpurun_out/c5_stream_latency.log
