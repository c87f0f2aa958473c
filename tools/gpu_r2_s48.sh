#!/bin/bash
# lab: k_match7, how far the window moves at a time (the wait at a move is for walks of positions below the shift)
mkdir -p gpurun_out/s48
timeout 250 python tools/gpu_matchlab.py --mib 256 SZL_MATCH_KERNEL=2 SZL_MATCH_KERNEL=4,SZL_SHIFT=12288 SZL_MATCH_KERNEL=4,SZL_SHIFT=8192 SZL_MATCH_KERNEL=4,SZL_SHIFT=4096 SZL_MATCH_KERNEL=4,SZL_SHIFT=2048 SZL_MATCH_KERNEL=4,SZL_SHIFT=2048,SZL_STRIPE_KIB=1024 SZL_MATCH_KERNEL=4,SZL_SHIFT=4096,SZL_SLICE=64 > gpurun_out/s48/enwik_256.log 2>&1
timeout 120 python tools/gpu_matchlab.py --mib 128 --kind logs --level 9 SZL_MATCH_KERNEL=2 SZL_MATCH_KERNEL=4,SZL_SHIFT=4096 SZL_MATCH_KERNEL=4,SZL_SHIFT=2048 > gpurun_out/s48/logs_l9.log 2>&1
cat gpurun_out/s48/*.log | grep -v amdgpu
