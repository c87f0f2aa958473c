#!/bin/bash
# round 2 session 11: four-context stage-B engine, first run + sweep
mkdir -p gpurun_out/s11
timeout 200 python tools/gpu_matchlab.py --mib 64 --oracle SZL_NCTX=2 SZL_NCTX=4 > gpurun_out/s11/first.log 2>&1
echo "rc=$?" >> gpurun_out/s11/first.log
cat gpurun_out/s11/first.log
if grep -q "DIFFERS\|rc=124\|Error\|error" gpurun_out/s11/first.log; then exit 0; fi
timeout 600 python tools/gpu_matchlab.py --mib 256 SZL_NCTX=2 SZL_NCTX=4 \
  SZL_NCTX=4,SZL_FTH4=48,SZL_VTH4=24,SZL_QKEEP4=96,SZL_VKEEP4=8 \
  SZL_NCTX=4,SZL_FTH4=96,SZL_VTH4=24,SZL_QKEEP4=96,SZL_VKEEP4=8 \
  SZL_NCTX=4,SZL_FTH4=64,SZL_VTH4=16,SZL_QKEEP4=96,SZL_VKEEP4=8 \
  SZL_NCTX=4,SZL_FTH4=64,SZL_VTH4=40,SZL_QKEEP4=96,SZL_VKEEP4=8 \
  SZL_NCTX=4,SZL_FTH4=64,SZL_VTH4=24,SZL_QKEEP4=64,SZL_VKEEP4=8 \
  SZL_NCTX=4,SZL_FTH4=64,SZL_VTH4=24,SZL_QKEEP4=128,SZL_VKEEP4=8 \
  SZL_NCTX=4,SZL_FTH4=64,SZL_VTH4=24,SZL_QKEEP4=96,SZL_VKEEP4=2 \
  SZL_NCTX=4,SZL_FTH4=64,SZL_VTH4=24,SZL_QKEEP4=96,SZL_VKEEP4=20 \
  SZL_NCTX=4,SZL_FTH4=32,SZL_VTH4=32,SZL_QKEEP4=160,SZL_VKEEP4=8 \
  > gpurun_out/s11/sweep.log 2>&1
cat gpurun_out/s11/sweep.log
timeout 300 python tools/gpu_matchlab.py --mib 128 --kind logs --level 9 SZL_NCTX=2 SZL_NCTX=4 > gpurun_out/s11/logs9.log 2>&1
cat gpurun_out/s11/logs9.log
