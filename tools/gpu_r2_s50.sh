#!/bin/bash
# lab: k_match8 with chunk staging that keeps up (six 16-byte loads in flight per lane, up to three chunks per lock hold) + loop counters
mkdir -p gpurun_out/s50
timeout 90 python tools/gpu_matchlab.py --mib 32 --oracle SZL_MATCH_KERNEL=5 SZL_MATCH_KERNEL=5,SZL_STRIPE_MIN=1,SZL_STRIPE_KIB=1024 > gpurun_out/s50/enwik_l6_oracle.log 2>&1
echo "rc $?" >> gpurun_out/s50/enwik_l6_oracle.log
grep -v amdgpu gpurun_out/s50/enwik_l6_oracle.log
if grep -q "rc 0" gpurun_out/s50/enwik_l6_oracle.log; then
timeout 200 python tools/gpu_matchlab.py --mib 256 SZL_MATCH_KERNEL=2 SZL_MATCH_KERNEL=5 SZL_MATCH_KERNEL=5,SZL_LOWWATER=3072 SZL_MATCH_KERNEL=5,SZL_LOWWATER=12288 SZL_MATCH_KERNEL=5,SZL_FTH2=48 SZL_MATCH_KERNEL=5,SZL_STRIPE_KIB=1024 > gpurun_out/s50/enwik_256.log 2>&1
timeout 100 python tools/gpu_matchlab.py --mib 128 --debug --reps 1 SZL_MATCH_KERNEL=5 2>&1 | grep -v "stage B\|match:" > gpurun_out/s50/enwik_dbg.log
timeout 100 python tools/gpu_matchlab.py --mib 128 --kind logs --level 9 SZL_MATCH_KERNEL=2 SZL_MATCH_KERNEL=5 > gpurun_out/s50/logs_l9.log 2>&1
grep -v amdgpu gpurun_out/s50/enwik_256.log gpurun_out/s50/enwik_dbg.log gpurun_out/s50/logs_l9.log
fi
