#!/bin/bash
# round 4, GPU call 3: the default bench line with the full-size configs, then the whole GPU suite
mkdir -p gpurun_out; cd "$(dirname "$0")/.."
( time timeout 900 python bench.py --steps 10 --warmup 2 ) > gpurun_out/c3_bench.json 2> gpurun_out/c3_bench.err
echo "bench rc $?" >> gpurun_out/c3_bench.err
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_inflate_fuzz.py::test_code_sets_the_reference_table_decodes_differently > gpurun_out/c3_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/c3_pytest.log
python - <<'PY'
import json
l=[x for x in open('gpurun_out/c3_bench.json') if x.startswith('{')]
if l:
    j=json.loads(l[-1]); print(j['value'], j['ms_per_step'], j['stage_ms'], j['roofline']); print(json.dumps(j.get('configs'),indent=1)[:6000]); print(j.get('cpu_baseline'), j.get('cpu_baseline_all_cores'))
PY
tail -n 5 gpurun_out/c3_bench.err; tail -n 12 gpurun_out/c3_pytest.log
