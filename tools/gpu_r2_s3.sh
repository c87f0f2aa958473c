#!/bin/bash
# Round 2, GPU session 3: k_match4 (two-context hand-written engine) — parity first, then the knob sweep.
mkdir -p gpurun_out/s3
export PYTHONUNBUFFERED=1
timeout 300 python tools/gpu_matchlab.py --mib 64 --oracle SZL_MATCH_KERNEL=1 SZL_MATCH_KERNEL=2 > gpurun_out/s3/lab_first.log 2>&1
cat gpurun_out/s3/lab_first.log
( time timeout 600 python -m pytest tests/test_gpu_deflate.py -m gpu -x -q ) > gpurun_out/s3/tests.log 2>&1
tail -5 gpurun_out/s3/tests.log
timeout 600 python tools/gpu_matchlab.py --mib 256 --oracle \
  SZL_MATCH_KERNEL=1 \
  SZL_MATCH_KERNEL=2,SZL_FTH2=32,SZL_VTH2=40,SZL_QKEEP=48,SZL_VKEEP=24 \
  SZL_MATCH_KERNEL=2,SZL_FTH2=32,SZL_VTH2=40,SZL_QKEEP=32,SZL_VKEEP=16 \
  SZL_MATCH_KERNEL=2,SZL_FTH2=32,SZL_VTH2=40,SZL_QKEEP=64,SZL_VKEEP=24 \
  SZL_MATCH_KERNEL=2,SZL_FTH2=24,SZL_VTH2=32,SZL_QKEEP=48,SZL_VKEEP=16 \
  SZL_MATCH_KERNEL=2,SZL_FTH2=48,SZL_VTH2=48,SZL_QKEEP=48,SZL_VKEEP=24 \
  SZL_MATCH_KERNEL=2,SZL_FTH2=32,SZL_VTH2=24,SZL_QKEEP=48,SZL_VKEEP=12 \
  SZL_MATCH_KERNEL=2,SZL_FTH2=32,SZL_VTH2=56,SZL_QKEEP=40,SZL_VKEEP=32 \
  SZL_MATCH_KERNEL=2,SZL_FTH2=64,SZL_VTH2=40,SZL_QKEEP=40,SZL_VKEEP=16 \
  SZL_MATCH_KERNEL=2,SZL_FTH2=16,SZL_VTH2=40,SZL_QKEEP=56,SZL_VKEEP=24 \
  > gpurun_out/s3/lab_sweep.log 2>&1
cat gpurun_out/s3/lab_sweep.log
timeout 300 python tools/gpu_matchlab.py --mib 128 --kind logs --level 9 --reps 2 SZL_MATCH_KERNEL=1 SZL_MATCH_KERNEL=2 > gpurun_out/s3/lab_logs.log 2>&1
cat gpurun_out/s3/lab_logs.log
true
