#!/bin/bash
# end-of-round check as the driver runs it: GPU tests, smoke(), bench line
mkdir -p gpurun_out/final
( time timeout 2400 python -m pytest tests -x -q -m gpu ) 2>&1 | tail -8 > gpurun_out/final/tests.log
cat gpurun_out/final/tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 > gpurun_out/final/smoke.log; cat gpurun_out/final/smoke.log
timeout 900 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err; tail -c 2500 gpurun_out/final/bench.json
