#!/bin/bash
# round 3, session b: first run of k_match5 (bucket order): parity against the oracle, then timing against k_match4
mkdir -p gpurun_out/r3b
timeout 120 python tools/gpu_matchlab.py --mib 8 --oracle SZL_MATCH_KERNEL=2 SZL_MATCH_KERNEL=5 > gpurun_out/r3b/m5_parity8.log 2>&1; grep -v amdgpu gpurun_out/r3b/m5_parity8.log
timeout 200 python tools/gpu_matchlab.py --mib 64 --oracle --debug SZL_MATCH_KERNEL=2 SZL_MATCH_KERNEL=5 > gpurun_out/r3b/m5_parity64.log 2>&1; grep -v amdgpu gpurun_out/r3b/m5_parity64.log
timeout 200 python tools/gpu_matchlab.py --mib 256 SZL_MATCH_KERNEL=2 SZL_MATCH_KERNEL=5 > gpurun_out/r3b/m5_256.log 2>&1; grep -v amdgpu gpurun_out/r3b/m5_256.log
timeout 200 python tools/gpu_matchlab.py --mib 64 --kind logs --level 9 --oracle SZL_MATCH_KERNEL=2 SZL_MATCH_KERNEL=5 > gpurun_out/r3b/m5_logs9.log 2>&1; grep -v amdgpu gpurun_out/r3b/m5_logs9.log
