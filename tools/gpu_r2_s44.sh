#!/bin/bash
# lab: per-kernel times of the chain-compression form (k_links4 vs k_match6)
mkdir -p gpurun_out/s44
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/s44 -o k3 -- python $GRAFT_REPO_ROOT/tools/gpu_matchlab.py --mib 256 SZL_MATCH_KERNEL=3 > $GRAFT_REPO_ROOT/gpurun_out/s44/lab.log 2> $GRAFT_REPO_ROOT/gpurun_out/s44/rocprof.err
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/s44 -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-200
grep -v amdgpu gpurun_out/s44/lab.log
find gpurun_out/s44 -name "*kernel_trace.csv" -delete
