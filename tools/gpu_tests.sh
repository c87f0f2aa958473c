#!/bin/bash
# Run on the GPU box: parity tests + smoke + bench; logs under gpurun_out/
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -15
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
bash tools/gpu_two_device_check.sh 2>&1 | tail -8        # (a no-op on a one-GPU box)
