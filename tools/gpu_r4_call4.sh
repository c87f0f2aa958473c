#!/bin/bash
# round 4, GPU call 4: the streaming Inflater's long-input path (tests + InflaterInputStream by buffer size), stage-B threshold sweep on the tail program
mkdir -p gpurun_out; cd "$(dirname "$0")/.."
timeout 600 python -m pytest tests/test_gpu_inflate_stream_bulk.py tests/test_gpu_inflate.py tests/test_gpu_gzip.py -m gpu -q > gpurun_out/c4_bulk.log 2>&1
echo "bulk rc $?" >> gpurun_out/c4_bulk.log
timeout 300 python tools/gpu_stream_latency.py --entries 300 > gpurun_out/c4_stream_latency.log 2>&1
echo "latency rc $?" >> gpurun_out/c4_stream_latency.log
timeout 400 python tools/gpu_matchlab.py --product --mib 256 --oracle SZL_B9=1 "SZL9_FTH=16" "SZL9_VTHT1=2" "SZL9_VTHT1=4" "SZL9_KTAIL1=2" "SZL9_QKEEP=48" "SZL9_QKEEP=80" "SZL9_VTH=3" "SZL9_VTH=1" "SZL_SLICE=192" "SZL_SLICE=96" "SZL9_FTH=16,SZL9_QKEEP=56" "SZL9_FTH=20,SZL9_VTH=3" "SZL9_MTH=56" > gpurun_out/c4_lab.log 2>&1
tail -n 25 gpurun_out/c4_bulk.log; tail -n 12 gpurun_out/c4_stream_latency.log; tail -n 16 gpurun_out/c4_lab.log
