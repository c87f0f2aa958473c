#!/bin/bash
# Run on the GPU box: bench line + rocprofv3 kernel stats of the same command
mkdir -p gpurun_out/prof
true
true
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof/bench_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof/rocprof.err
cd $GRAFT_REPO_ROOT
ls -R gpurun_out/prof | head -20
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); echo $f; head -25 "$f"
