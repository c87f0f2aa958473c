#!/bin/bash
# per-kernel stats on 1 GiB enwik L6 for the C-stage variants
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  SZL_CWIN=$v rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_cwin$v -- python $GRAFT_REPO_ROOT/tools/gpu_scale.py 1024 > /dev/null 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_cwin$v -name "*kernel_stats.csv" | head -1)
  echo "== SZL_CWIN=$v"; head -20 "$f" | cut -d, -f1-6
done
