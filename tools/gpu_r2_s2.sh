#!/bin/bash
# Round 2, GPU session 2: k_match3 (hand-written QUICK/VERIFY loops) — parity on the deflate suite, then the knob sweep.
mkdir -p gpurun_out/s2
export PYTHONUNBUFFERED=1
python tools/gpu_matchlab.py --mib 64 --oracle SZL_MATCH_KERNEL=1 SZL_MATCH_KERNEL=2,SZL_VMODE=0 SZL_MATCH_KERNEL=2,SZL_VMODE=1 > gpurun_out/s2/lab_first.log 2>&1
cat gpurun_out/s2/lab_first.log
( time python -m pytest tests/test_gpu_deflate.py -m gpu -x -q ) > gpurun_out/s2/tests.log 2>&1
tail -5 gpurun_out/s2/tests.log
python tools/gpu_matchlab.py --mib 256 --debug --reps 1 SZL_MATCH_KERNEL=2,SZL_VMODE=1 SZL_MATCH_KERNEL=2,SZL_VMODE=0 > gpurun_out/s2/lab_debug.log 2>&1
cat gpurun_out/s2/lab_debug.log
python tools/gpu_matchlab.py --mib 256 --oracle \
  SZL_MATCH_KERNEL=1 \
  SZL_MATCH_KERNEL=2,SZL_VMODE=0,SZL_FTH2=16,SZL_VTH2=20,SZL_QKEEP=24,SZL_VKEEP=12 \
  SZL_MATCH_KERNEL=2,SZL_VMODE=1,SZL_FTH2=16,SZL_VTH2=20,SZL_QKEEP=24,SZL_VKEEP=12 \
  SZL_MATCH_KERNEL=2,SZL_VMODE=1,SZL_FTH2=16,SZL_VTH2=20,SZL_QKEEP=16,SZL_VKEEP=8 \
  SZL_MATCH_KERNEL=2,SZL_VMODE=1,SZL_FTH2=16,SZL_VTH2=20,SZL_QKEEP=32,SZL_VKEEP=12 \
  SZL_MATCH_KERNEL=2,SZL_VMODE=1,SZL_FTH2=12,SZL_VTH2=16,SZL_QKEEP=20,SZL_VKEEP=8 \
  SZL_MATCH_KERNEL=2,SZL_VMODE=1,SZL_FTH2=24,SZL_VTH2=24,SZL_QKEEP=16,SZL_VKEEP=8 \
  SZL_MATCH_KERNEL=2,SZL_VMODE=1,SZL_FTH2=20,SZL_VTH2=16,SZL_QKEEP=24,SZL_VKEEP=6 \
  SZL_MATCH_KERNEL=2,SZL_VMODE=1,SZL_FTH2=16,SZL_VTH2=28,SZL_QKEEP=24,SZL_VKEEP=16 \
  > gpurun_out/s2/lab_sweep.log 2>&1
cat gpurun_out/s2/lab_sweep.log
python tools/gpu_matchlab.py --mib 128 --kind logs --level 9 --reps 2 SZL_MATCH_KERNEL=1 SZL_MATCH_KERNEL=2,SZL_VMODE=1 > gpurun_out/s2/lab_logs.log 2>&1
cat gpurun_out/s2/lab_logs.log
true
