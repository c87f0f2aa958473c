#!/bin/bash
python -m pytest tests/test_gpu_deflate.py -x -q -m gpu 2>&1 | tail -2
for cfg in "16 20" "24 24" "8 16" "16 32" "32 32" "24 12"; do
  set -- $cfg
  echo "FTH=$1 VTH=$2: $(SZL_FTH=$1 SZL_VTH=$2 python tools/gpu_scale.py 256 2>&1 | grep 'n=256MiB' | tail -1 | sed 's/.*\[ck/[ck/')"
done
