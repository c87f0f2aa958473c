#!/bin/bash
mkdir -p gpurun_out/s38
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/s38/bench.json 2> gpurun_out/s38/bench.err
python -c "
import json; d=json.loads(open('gpurun_out/s38/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['stage_ms'], d['roofline']['frac'], d['parity'])"
SZL_C3_N=20000 timeout 600 python tools/gpu_configs.py c3 2>&1 | grep "c3:" | tee gpurun_out/s38/c3.log
timeout 1500 python -m pytest tests/test_gpu_deflate.py tests/test_gpu_headline.py tests/test_gpu_window.py tests/test_gpu_multi.py -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/s38/tests.log
