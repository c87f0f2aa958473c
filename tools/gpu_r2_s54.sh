#!/bin/bash
# tiles cut into equal parts per segment: parity subset + config 3
mkdir -p gpurun_out/s54
timeout 45 python -m pytest tests/test_gpu_stage_b_forms.py tests/test_gpu_deflate.py -m gpu -x -q -k "forms or boundary_sizes or many_small or randomised or batch_bit_exact or golden" 2>&1 | tail -4 > gpurun_out/s54/pytest.log; cat gpurun_out/s54/pytest.log
SZL_C3_N=50000 timeout 30 python tools/gpu_configs.py c3 2>&1 | grep "^c3: \|spot" > gpurun_out/s54/c3.log; cat gpurun_out/s54/c3.log
