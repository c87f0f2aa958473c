#!/bin/bash
for v in 8 32; do
for f in 16 4; do
echo "== SZL_STRIDE=$v SZL_FTH=$f"; SZL_FTH=$f SZL_STRIDE=$v SZL_DEBUG=1 python tools/gpu_scale.py 1024 2>&1 | grep -v "^gen" | grep -v "quick wave" | tail -3 | head -2
done; done
echo "== SZL_STRIDE=16 SZL_FTH=4"; SZL_FTH=4 SZL_DEBUG=1 python tools/gpu_scale.py 1024 2>&1 | grep -v "^gen" | grep -v "quick wave" | tail -3 | head -2
