#!/bin/bash
# round 2 session 16: pipelined stage A (k_links3)
mkdir -p gpurun_out/s16
timeout 300 python tools/gpu_matchlab.py --mib 256 --oracle SZL_LINKS=2 SZL_LINKS=3 > gpurun_out/s16/links.log 2>&1
echo "rc=$?" >> gpurun_out/s16/links.log
cat gpurun_out/s16/links.log
if grep -q "DIFFERS\|rc=124\|Error\|error" gpurun_out/s16/links.log; then exit 0; fi
timeout 900 python -m pytest tests/test_gpu_deflate.py tests/test_gpu_window.py tests/test_gpu_inflate.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/s16/tests.log
cat gpurun_out/s16/tests.log
timeout 300 python bench.py --steps 5 --warmup 1 > gpurun_out/s16/bench.json 2> gpurun_out/s16/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/s16/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["stage_ms"], d["parity"])
PY
