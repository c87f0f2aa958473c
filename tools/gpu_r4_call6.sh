#!/bin/bash
# round 4, GPU call 6: the whole GPU suite on the current tree, then everything profiles/r04 holds for the headline command
mkdir -p gpurun_out; cd "$(dirname "$0")/.."
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/c6_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/c6_pytest.log
timeout 1500 bash tools/gpu_profile_round.sh > gpurun_out/c6_profile.log 2>&1
echo "profile rc $?" >> gpurun_out/c6_profile.log
tail -n 6 gpurun_out/c6_pytest.log; tail -n 60 gpurun_out/c6_profile.log
