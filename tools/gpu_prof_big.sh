#!/bin/bash
mkdir -p gpurun_out/profbig
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/profbig -o big -- python $GRAFT_REPO_ROOT/tools/gpu_big.py 2.5 9 > $GRAFT_REPO_ROOT/gpurun_out/profbig/out.txt 2>&1
cd $GRAFT_REPO_ROOT
cut -d, -f1-4 gpurun_out/profbig/big_kernel_stats.csv | cut -c1-150 | head -14
