#!/bin/bash
# lab: stage B fed from a ring (k_match8, SZL_MATCH_KERNEL=5) against the oracle and against k_match4
mkdir -p gpurun_out/s49
timeout 90 python tools/gpu_matchlab.py --mib 32 --oracle SZL_MATCH_KERNEL=2 SZL_MATCH_KERNEL=5 SZL_MATCH_KERNEL=5,SZL_STRIPE_MIN=1,SZL_STRIPE_KIB=1024 > gpurun_out/s49/enwik_l6_oracle.log 2>&1
echo "rc $?" >> gpurun_out/s49/enwik_l6_oracle.log
grep -v amdgpu gpurun_out/s49/enwik_l6_oracle.log
if grep -q "rc 0" gpurun_out/s49/enwik_l6_oracle.log; then
timeout 90 python tools/gpu_matchlab.py --mib 24 --kind logs --level 9 --oracle SZL_MATCH_KERNEL=5,SZL_STRIPE_MIN=1 > gpurun_out/s49/logs_l9_oracle.log 2>&1
timeout 200 python tools/gpu_matchlab.py --mib 256 SZL_MATCH_KERNEL=2 SZL_MATCH_KERNEL=5 SZL_MATCH_KERNEL=5,SZL_STRIPE_KIB=1024 SZL_MATCH_KERNEL=5,SZL_LOWWATER=3072 SZL_MATCH_KERNEL=5,SZL_LOWWATER=10240 SZL_MATCH_KERNEL=5,SZL_SLICE=64 SZL_MATCH_KERNEL=5,SZL_SLICE=256 SZL_MATCH_KERNEL=5,SZL_FTH2=16 SZL_MATCH_KERNEL=5,SZL_FTH2=48 > gpurun_out/s49/enwik_256.log 2>&1
grep -v amdgpu gpurun_out/s49/logs_l9_oracle.log gpurun_out/s49/enwik_256.log
fi
