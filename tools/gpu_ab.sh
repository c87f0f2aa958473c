#!/bin/bash
# A/B of the stage-A variants on 1 GiB enwik L6 (+ parity on 64 MiB for each variant).
for v in 2 1; do
echo "== SZL_LINKS=$v"; SZL_LINKS=$v python tools/gpu_scale.py 64 1024 2>&1 | grep -v "^gen"
done
