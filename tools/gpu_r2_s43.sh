#!/bin/bash
# lab: chain compression (k_links4 + k_match6, SZL_MATCH_KERNEL=3) against the oracle and against k_match4
mkdir -p gpurun_out/s43
timeout 150 python tools/gpu_matchlab.py --mib 32 --oracle SZL_MATCH_KERNEL=2 SZL_MATCH_KERNEL=3 > gpurun_out/s43/enwik_l6_oracle.log 2>&1
timeout 150 python tools/gpu_matchlab.py --mib 32 --kind logs --level 5 --oracle SZL_MATCH_KERNEL=3 > gpurun_out/s43/logs_l5_oracle.log 2>&1
timeout 150 python tools/gpu_matchlab.py --mib 32 --kind dickens --level 6 --oracle SZL_MATCH_KERNEL=3 > gpurun_out/s43/dickens_l6_oracle.log 2>&1
timeout 200 python tools/gpu_matchlab.py --mib 256 SZL_MATCH_KERNEL=2 SZL_MATCH_KERNEL=3 SZL_MATCH_KERNEL=3,SZL_STH6=2 SZL_MATCH_KERNEL=3,SZL_STH6=24 SZL_MATCH_KERNEL=3,SZL_VTH6=4 SZL_MATCH_KERNEL=3,SZL_FTH6=16 > gpurun_out/s43/enwik_256.log 2>&1
timeout 150 python tools/gpu_matchlab.py --mib 128 --debug SZL_MATCH_KERNEL=3 > gpurun_out/s43/enwik_dbg.log 2>&1
cat gpurun_out/s43/*.log | grep -v amdgpu
