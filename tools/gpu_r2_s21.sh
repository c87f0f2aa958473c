#!/bin/bash
# lab: scan_end filter as ONE unaligned 16-bit LDS read instead of two byte reads (separate build of szl_kernels_match2.hip)
mkdir -p gpurun_out/s21
timeout 300 python tools/gpu_matchlab.py --mib 256 SZL_MATCH_KERNEL=2 > gpurun_out/s21/base.log 2>&1
cp sharpziplib_amd/csrc/lab_u16.so sharpziplib_amd/csrc/libszl_amd.so
timeout 300 python tools/gpu_matchlab.py --mib 64 --oracle SZL_MATCH_KERNEL=2 > gpurun_out/s21/u16_oracle.log 2>&1
timeout 300 python tools/gpu_matchlab.py --mib 256 SZL_MATCH_KERNEL=2 SZL_VTH2=12 SZL_VTH2=6 SZL_QKEEP=40 SZL_QKEEP=56 > gpurun_out/s21/u16.log 2>&1
timeout 300 python tools/gpu_matchlab.py --mib 128 --kind logs --level 9 SZL_MATCH_KERNEL=2 > gpurun_out/s21/u16_logs.log 2>&1
cat gpurun_out/s21/base.log gpurun_out/s21/u16_oracle.log gpurun_out/s21/u16.log gpurun_out/s21/u16_logs.log | grep -v amdgpu
