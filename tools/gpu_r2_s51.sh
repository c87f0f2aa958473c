#!/bin/bash
# k_match4 with tiles as long as LDS allows (21504 positions) against 16384; the parity tests of the three forms of the full search
mkdir -p gpurun_out/s51
timeout 90 python tools/gpu_matchlab.py --mib 32 --oracle SZL_MATCH_KERNEL=2 > gpurun_out/s51/enwik_l6_oracle.log 2>&1
timeout 200 python tools/gpu_matchlab.py --mib 256 SZL_MATCH_KERNEL=2,SZL_TILE_LEN=16384 SZL_MATCH_KERNEL=2 SZL_MATCH_KERNEL=2,SZL_TILE_LEN=20480 SZL_MATCH_KERNEL=2,SZL_TILE_LEN=18432 > gpurun_out/s51/enwik_256.log 2>&1
timeout 100 python tools/gpu_matchlab.py --mib 128 --kind logs --level 9 SZL_MATCH_KERNEL=2,SZL_TILE_LEN=16384 SZL_MATCH_KERNEL=2 > gpurun_out/s51/logs_l9.log 2>&1
grep -hv amdgpu gpurun_out/s51/enwik_l6_oracle.log gpurun_out/s51/enwik_256.log gpurun_out/s51/logs_l9.log
timeout 300 python -m pytest tests/test_gpu_stage_b_forms.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/s51/pytest_forms.log
cat gpurun_out/s51/pytest_forms.log
