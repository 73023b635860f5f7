#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_golden.py -m gpu -q -x -k "vram or streamed or reproducible" 2>&1 | tail -8
timeout 1500 python -m pytest tests/test_fullsize.py -m gpu -q -x -s 2>&1 | grep -E "full size|passed|failed|Error" | tail -8
