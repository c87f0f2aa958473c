#!/bin/bash
# A/B of the C-stage window kernels on 1 GiB enwik L6 (+ parity on 64 MiB for each variant).
for v in 16 8; do
echo "== SZL_CWIN=$v"; SZL_CWIN=$v python tools/gpu_scale.py 1024 2>&1 | grep -v "^gen"
done
