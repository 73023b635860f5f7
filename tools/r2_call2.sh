#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
lscpu | grep -E "Model name|^CPU\(s\)" > gpurun_out/hostcpu.txt; cat gpurun_out/hostcpu.txt
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu_r2b.log 2>&1; tail -15 gpurun_out/pytest_gpu_r2b.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_r2b.log 2>&1; tail -3 gpurun_out/smoke_r2b.log
