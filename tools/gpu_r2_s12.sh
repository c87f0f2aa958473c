#!/bin/bash
mkdir -p gpurun_out/s12
timeout 300 python tools/gpu_matchlab.py --mib 128 --debug --reps 1 SZL_NCTX=2 SZL_NCTX=2,SZL_VTH2=32,SZL_VKEEP=8 SZL_NCTX=2,SZL_QKEEP=80 > gpurun_out/s12/counters.log 2>&1
cat gpurun_out/s12/counters.log
