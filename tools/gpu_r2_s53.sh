#!/bin/bash
# config 3 (many 64 KiB entries in one call): do the longer stage-B tiles (3 x 21504 + 1024 per entry instead of 4 x 16384) cost anything?
mkdir -p gpurun_out/s53
SZL_C3_N=50000 SZL_TILE_LEN=16384 timeout 70 python tools/gpu_configs.py c3 2>&1 | grep "^c3" > gpurun_out/s53/c3_tiles16k.log
SZL_C3_N=50000 timeout 70 python tools/gpu_configs.py c3 2>&1 | grep "^c3" > gpurun_out/s53/c3_default.log
echo "# SZL_TILE_LEN=16384"; cat gpurun_out/s53/c3_tiles16k.log; echo "# default (21504)"; cat gpurun_out/s53/c3_default.log
