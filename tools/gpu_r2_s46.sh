#!/bin/bash
# lab: what the 8 KiB tiles and the fetch frequency cost (k_match4 with 8 KiB tiles; k_match6 with rarer fetches)
mkdir -p gpurun_out/s46
timeout 300 python tools/gpu_matchlab.py --mib 256 SZL_MATCH_KERNEL=2 SZL_MATCH_KERNEL=2,SZL_TILE_LEN=8192 SZL_MATCH_KERNEL=2,SZL_TILE_LEN=4096 SZL_MATCH_KERNEL=3 SZL_MATCH_KERNEL=3,SZL_FTH6=64 SZL_MATCH_KERNEL=3,SZL_FTH6=96 SZL_MATCH_KERNEL=3,SZL_SLICE6=256 SZL_MATCH_KERNEL=3,SZL_TILE_LEN=4096 SZL_MATCH_KERNEL=3,SZL_QKEEP6=48,SZL_VTH6=4 > gpurun_out/s46/lab256.log 2>&1
grep -v amdgpu gpurun_out/s46/lab256.log
