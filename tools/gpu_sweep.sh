#!/bin/bash
# stage-B scheduler thresholds (full search) on 512 MiB enwik L6
for cfg in "16 20" "8 20" "12 20" "24 20" "16 12" "16 28" "12 14" "8 12" "20 24" "12 28"; do
  set -- $cfg
  echo "FTH=$1 VTH=$2: $(SZL_MATCH_MODE=0 SZL_FTH=$1 SZL_VTH=$2 python tools/gpu_scale.py 512 2>&1 | grep 'n=512MiB' | tail -1 | sed 's/.*\[ck/[ck/' | cut -c1-60)"
done
