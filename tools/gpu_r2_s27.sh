#!/bin/bash
mkdir -p gpurun_out/s27
for ck in 128 256 512; do
  echo "== SZL_INF_CHUNK_KIB=$ck"
  SZL_INF_CHUNK_KIB=$ck SZL_DEBUG=1 timeout 600 python tools/gpu_inflate_perf.py 1024 2 2>&1 | grep -v "amdgpu\|match\|stage B" | grep "inflate par\|single" | head -8
done > gpurun_out/s27/chunks_enwik.log 2>&1
cat gpurun_out/s27/chunks_enwik.log
