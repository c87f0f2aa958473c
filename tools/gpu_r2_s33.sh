#!/bin/bash
mkdir -p gpurun_out/s33
for r in 4096 2048 1024 8192; do
  echo "== SZL_RANGE_LEN=$r"; SZL_RANGE_LEN=$r timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['stage_ms'])"
done > gpurun_out/s33/range.log 2>&1
cat gpurun_out/s33/range.log
