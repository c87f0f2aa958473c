#!/bin/bash
# First GPU call of round 3: the two lab forms the timing model points at, each under its own timeout.
#   1. k_match4o (two-pass hand-out; never run on a device): parity first, then time — also with fewer waves per workgroup
#      (SZL_ORDER_TH=70000 makes every position first-class, i.e. plain k_match4 order with the 64-at-a-time hand-out);
#   2. the ring (k_match8) with fewer waves per workgroup: the model says 50 -> 38..43 ms per GiB at 8 waves.
mkdir -p gpurun_out/r3a
SZL_TEST_UNVALIDATED=1 timeout 150 python -m pytest tests/test_ordered_handout.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r3a/ordered_parity.log; cat gpurun_out/r3a/ordered_parity.log
if grep -q " passed" gpurun_out/r3a/ordered_parity.log && ! grep -q "failed\|error" gpurun_out/r3a/ordered_parity.log; then
timeout 240 python tools/gpu_matchlab.py --mib 256 SZL_MATCH_KERNEL=2 SZL_MATCH_KERNEL=2,SZL_ORDERED=1 SZL_MATCH_KERNEL=2,SZL_ORDERED=1,SZL_ORDER_TH=1024 \
  SZL_MATCH_KERNEL=2,SZL_ORDERED=1,SZL_ORDER_TH=70000 SZL_MATCH_KERNEL=2,SZL_ORDERED=1,SZL_ORDER_TH=70000,SZL_M4_WAVES=8 SZL_MATCH_KERNEL=2,SZL_ORDERED=1,SZL_ORDER_TH=70000,SZL_M4_WAVES=12 \
  SZL_MATCH_KERNEL=2,SZL_ORDERED=1,SZL_M4_WAVES=8 > gpurun_out/r3a/ordered_256.log 2>&1
grep -v amdgpu gpurun_out/r3a/ordered_256.log
fi
timeout 90 python tools/gpu_matchlab.py --mib 32 --oracle SZL_MATCH_KERNEL=4,SZL_RING_WAVES=8 SZL_MATCH_KERNEL=4,SZL_RING_WAVES=12,SZL_STRIPE_MIN=1,SZL_STRIPE_KIB=1024 > gpurun_out/r3a/ring_waves_oracle.log 2>&1
grep -v amdgpu gpurun_out/r3a/ring_waves_oracle.log
timeout 240 python tools/gpu_matchlab.py --mib 256 SZL_MATCH_KERNEL=2 SZL_MATCH_KERNEL=4 SZL_MATCH_KERNEL=4,SZL_RING_WAVES=12 SZL_MATCH_KERNEL=4,SZL_RING_WAVES=10 SZL_MATCH_KERNEL=4,SZL_RING_WAVES=8 \
  SZL_MATCH_KERNEL=4,SZL_RING_WAVES=6 SZL_MATCH_KERNEL=4,SZL_RING_WAVES=8,SZL_FTH2=16 SZL_MATCH_KERNEL=4,SZL_RING_WAVES=8,SZL_LOWWATER=3072 > gpurun_out/r3a/ring_waves_256.log 2>&1
grep -v amdgpu gpurun_out/r3a/ring_waves_256.log
