#!/bin/bash
# lab: chain compression without the slow routine (first candidate = first chain element with the same three bytes), links4 out of LDS
mkdir -p gpurun_out/s45
timeout 150 python tools/gpu_matchlab.py --mib 32 --oracle SZL_MATCH_KERNEL=2 SZL_MATCH_KERNEL=3 SZL_MATCH_KERNEL=3,SZL_LINKS4=0 > gpurun_out/s45/enwik_l6_oracle.log 2>&1
timeout 150 python tools/gpu_matchlab.py --mib 32 --kind logs --level 5 --oracle SZL_MATCH_KERNEL=3 SZL_MATCH_KERNEL=3,SZL_LINKS4=0 > gpurun_out/s45/logs_l5_oracle.log 2>&1
timeout 150 python tools/gpu_matchlab.py --mib 24 --kind dickens --level 6 --oracle SZL_MATCH_KERNEL=3 > gpurun_out/s45/dickens_l6_oracle.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/s45 -o k3 -- python $GRAFT_REPO_ROOT/tools/gpu_matchlab.py --mib 256 SZL_MATCH_KERNEL=3 SZL_MATCH_KERNEL=3,SZL_L4_REFILL=32 SZL_MATCH_KERNEL=3,SZL_L4_REFILL=8 SZL_MATCH_KERNEL=2 > $GRAFT_REPO_ROOT/gpurun_out/s45/lab256.log 2> $GRAFT_REPO_ROOT/gpurun_out/s45/rocprof.err
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/s45 -name "*kernel_stats.csv" | head -1); head -6 "$f" | sed 's/(.*)"/"/' | cut -c1-160
cat gpurun_out/s45/*.log | grep -v amdgpu
find gpurun_out/s45 -name "*kernel_trace.csv" -delete
