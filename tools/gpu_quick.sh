#!/bin/bash
python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python tools/gpu_scale.py 64 1024 2>&1 | grep -v "^gen"
