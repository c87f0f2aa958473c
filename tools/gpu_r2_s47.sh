#!/bin/bash
# lab: stage B over stripes with a sliding window (k_match7, SZL_MATCH_KERNEL=4) against the oracle and against k_match4
mkdir -p gpurun_out/s47
timeout 120 python tools/gpu_matchlab.py --mib 32 --oracle SZL_MATCH_KERNEL=2 SZL_MATCH_KERNEL=4 SZL_MATCH_KERNEL=4,SZL_STRIPE_MIN=1,SZL_STRIPE_KIB=1024 > gpurun_out/s47/enwik_l6_oracle.log 2>&1
timeout 120 python tools/gpu_matchlab.py --mib 24 --kind logs --level 9 --oracle SZL_MATCH_KERNEL=2 SZL_MATCH_KERNEL=4,SZL_STRIPE_MIN=1 > gpurun_out/s47/logs_l9_oracle.log 2>&1
timeout 200 python tools/gpu_matchlab.py --mib 256 SZL_MATCH_KERNEL=2 SZL_MATCH_KERNEL=4 SZL_MATCH_KERNEL=4,SZL_STRIPE_KIB=128 SZL_MATCH_KERNEL=4,SZL_STRIPE_KIB=512 SZL_MATCH_KERNEL=4,SZL_STRIPE_KIB=1024 SZL_MATCH_KERNEL=4,SZL_SLICE=256 SZL_MATCH_KERNEL=4,SZL_FTH2=48 > gpurun_out/s47/enwik_256.log 2>&1
cat gpurun_out/s47/*.log | grep -v amdgpu
SZL_MATCH_KERNEL=4 SZL_STRIPE_MIN=1 SZL_STRIPE_KIB=64 timeout 400 python -m pytest tests/test_gpu_deflate.py tests/test_gpu_window.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/s47/pytest_k4.log
cat gpurun_out/s47/pytest_k4.log
