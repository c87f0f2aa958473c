#!/bin/bash
# lab: one-byte scan_end filter under the new thresholds (separate build of szl_kernels_match2.hip, -DSZL_LAB_FILTER1)
mkdir -p gpurun_out/s42
timeout 300 python tools/gpu_matchlab.py --mib 256 SZL_MATCH_KERNEL=2 > gpurun_out/s42/two_byte.log 2>&1
cp sharpziplib_amd/csrc/lab_f1.so sharpziplib_amd/csrc/libszl_amd.so
timeout 300 python tools/gpu_matchlab.py --mib 64 --oracle SZL_MATCH_KERNEL=2 > gpurun_out/s42/one_byte_oracle.log 2>&1
timeout 300 python tools/gpu_matchlab.py --mib 256 SZL_MATCH_KERNEL=2 SZL_VTH2=4 SZL_VTH2=1 SZL_QKEEP=56 > gpurun_out/s42/one_byte.log 2>&1
timeout 300 python tools/gpu_matchlab.py --mib 128 --kind logs --level 9 SZL_MATCH_KERNEL=2 > gpurun_out/s42/one_byte_logs.log 2>&1
cat gpurun_out/s42/two_byte.log gpurun_out/s42/one_byte_oracle.log gpurun_out/s42/one_byte.log gpurun_out/s42/one_byte_logs.log | grep -v amdgpu
