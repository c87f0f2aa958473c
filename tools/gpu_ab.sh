#!/bin/bash
# A/B of the stage-B slot counts on 1 GiB enwik L6 (+ parity on 64 MiB for each variant).
for v in 2 3 4 1; do
echo "== SZL_MATCH_SLOTS=$v"; SZL_MATCH_SLOTS=$v python tools/gpu_scale.py 64 1024 2>&1 | grep -v "^gen" | grep -v roundtrip
done
