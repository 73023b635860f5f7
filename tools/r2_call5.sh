#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_qu8_gpu.py -q 2>&1 | tail -40 > gpurun_out/pytest_qu8_r2.log; cat gpurun_out/pytest_qu8_r2.log
