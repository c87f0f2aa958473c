#!/bin/bash
# round 4, GPU call 1: the parity corners (exact-table Inflater mode, Reset() stale bits, DeflateFast slide corner), stage-B A/B
# (k_match4 / k_match9 with and without the tail program), then the whole GPU suite on the new default
mkdir -p gpurun_out; cd "$(dirname "$0")/.."
timeout 600 python -m pytest tests/test_gpu_inflate_fuzz.py tests/test_gpu_reset_bits.py tests/test_gpu_fast_slide_corner.py -m gpu -q > gpurun_out/c1_parity.log 2>&1
echo "parity rc $?" >> gpurun_out/c1_parity.log
timeout 400 python tools/gpu_matchlab.py --product --mib 256 --oracle SZL_B9=0 SZL_B9=1 "SZL_B9=1,SZL9_TAILP=0" "SZL_B9=1,SZL9_MTH=-1" "SZL_B9=1,SZL9_KTAIL1=1" "SZL_B9=1,SZL9_KTAIL1=4" "SZL_B9=1,SZL9_KTAIL=1" "SZL_B9=1,SZL9_KTAIL=1,SZL9_KTAIL1=1" "SZL_B9=1,SZL9_MTH=48" "SZL_B9=1,SZL9_FTH=16" "SZL_B9=1,SZL9_FTH=16,SZL9_KTAIL=1" > gpurun_out/c1_lab.log 2>&1
echo "lab rc $?" >> gpurun_out/c1_lab.log
timeout 200 python tools/gpu_matchlab.py --product --mib 256 --debug --reps 1 SZL_B9=1 "SZL_B9=1,SZL9_TAILP=0" > gpurun_out/c1_dbg.log 2>&1
timeout 200 python tools/gpu_matchlab.py --product --mib 256 --kind logs --level 9 --reps 2 SZL_B9=0 SZL_B9=1 "SZL_B9=1,SZL9_TAILP=0" > gpurun_out/c1_logs9.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/c1_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/c1_pytest.log
tail -n 15 gpurun_out/c1_parity.log; tail -n 14 gpurun_out/c1_lab.log; grep "stage B tiles" gpurun_out/c1_dbg.log; tail -n 4 gpurun_out/c1_logs9.log; tail -n 8 gpurun_out/c1_pytest.log
