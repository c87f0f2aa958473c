#!/bin/bash
mkdir -p gpurun_out/s18
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --hip-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/s18/prof -- python $GRAFT_REPO_ROOT/tools/gpu_stream_latency.py --entries 300 > $GRAFT_REPO_ROOT/gpurun_out/s18/prof.log 2>&1
cd $GRAFT_REPO_ROOT
tail -3 gpurun_out/s18/prof.log
for f in $(find gpurun_out/s18/prof -name "*hip_api_stats.csv" -o -name "*kernel_stats.csv"); do echo "== $f"; head -30 $f | cut -c1-160; done
find gpurun_out/s18/prof -name "*trace.csv" -size +20M -delete
