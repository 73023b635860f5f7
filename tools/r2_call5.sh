#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_qu8_gpu.py -q -s -k "vae or calibration" 2>&1 > gpurun_out/pytest_qu8_r2.log; grep -E "^E|passed|failed|calibration:|uint8 pass" gpurun_out/pytest_qu8_r2.log | head -40
