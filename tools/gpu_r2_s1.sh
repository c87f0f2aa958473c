#!/bin/bash
# Round 2, GPU session 1: parity of the new code paths, stage-B diagnostics and knob sweep, PMC evidence of the r1 kernel.
mkdir -p gpurun_out/s1
export PYTHONUNBUFFERED=1
( time python -m pytest tests -m gpu -x -q --durations=12 ) > gpurun_out/s1/tests.log 2>&1
tail -25 gpurun_out/s1/tests.log
# diagnostics: true lane use per phase
python tools/gpu_matchlab.py --mib 256 --debug --reps 1 SZL_MATCH_KERNEL=1 SZL_MATCH_KERNEL=2,SZL_NCTX=1 SZL_MATCH_KERNEL=2,SZL_NCTX=2 > gpurun_out/s1/lab_debug.log 2>&1
cat gpurun_out/s1/lab_debug.log
# sweep
python tools/gpu_matchlab.py --mib 256 --oracle \
  SZL_MATCH_KERNEL=1 \
  SZL_MATCH_KERNEL=2,SZL_NCTX=1,SZL_FTH2=16,SZL_VTH2=20,SZL_QKEEP=24,SZL_VKEEP=12 \
  SZL_MATCH_KERNEL=2,SZL_NCTX=1,SZL_FTH2=16,SZL_VTH2=20,SZL_QKEEP=32,SZL_VKEEP=16 \
  SZL_MATCH_KERNEL=2,SZL_NCTX=1,SZL_FTH2=12,SZL_VTH2=16,SZL_QKEEP=40,SZL_VKEEP=16 \
  SZL_MATCH_KERNEL=2,SZL_NCTX=2,SZL_FTH2=16,SZL_VTH2=24,SZL_QKEEP=40,SZL_VKEEP=16 \
  SZL_MATCH_KERNEL=2,SZL_NCTX=2,SZL_FTH2=24,SZL_VTH2=32,SZL_QKEEP=48,SZL_VKEEP=24 \
  SZL_MATCH_KERNEL=2,SZL_NCTX=2,SZL_FTH2=32,SZL_VTH2=40,SZL_QKEEP=48,SZL_VKEEP=32 \
  SZL_MATCH_KERNEL=2,SZL_NCTX=2,SZL_FTH2=32,SZL_VTH2=32,SZL_QKEEP=32,SZL_VKEEP=24 \
  SZL_MATCH_KERNEL=2,SZL_NCTX=2,SZL_FTH2=24,SZL_VTH2=40,SZL_QKEEP=56,SZL_VKEEP=32 \
  SZL_MATCH_KERNEL=2,SZL_NCTX=2,SZL_FTH2=40,SZL_VTH2=48,SZL_QKEEP=40,SZL_VKEEP=40 \
  SZL_MATCH_KERNEL=2,SZL_NCTX=2,SZL_FTH2=16,SZL_VTH2=48,SZL_QKEEP=24,SZL_VKEEP=16 \
  > gpurun_out/s1/lab_sweep.log 2>&1
cat gpurun_out/s1/lab_sweep.log
# level 9 logs (config 5 shape), full search forced
python tools/gpu_matchlab.py --mib 128 --kind logs --level 9 --reps 2 SZL_MATCH_KERNEL=1 SZL_MATCH_KERNEL=2,SZL_NCTX=2,SZL_FTH2=16,SZL_VTH2=24,SZL_QKEEP=40,SZL_VKEEP=16 > gpurun_out/s1/lab_logs.log 2>&1
cat gpurun_out/s1/lab_logs.log
# PMC evidence for the round-1 kernel ("before")
export SZL_MATCH_KERNEL=1
bash tools/gpu_pmc.sh > gpurun_out/s1/pmc_k1.log 2>&1
tail -12 gpurun_out/s1/pmc_k1.log
mkdir -p gpurun_out/s1/pmc_k1 && cp gpurun_out/pmc/*.csv gpurun_out/s1/pmc_k1/ 2>/dev/null
true
