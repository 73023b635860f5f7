#!/bin/bash
# round 4, GPU call 13: the lean linear kernel -- kernel tests, per-shape probe, golden chain, full-size parity, bench A/B
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_linear_small.py tests/test_tblock_tail.py -m gpu -x -q > gpurun_out/r4c13_tests.log 2>&1; tail -5 gpurun_out/r4c13_tests.log
timeout 300 python tools/linear_small_probe.py > gpurun_out/r4c13_probe.log 2>&1; cat gpurun_out/r4c13_probe.log
timeout 600 python -m pytest tests/test_golden.py -m gpu -x -q -s -k "chains or transformer_block or unet_tiny" > gpurun_out/r4c13_golden.log 2>&1; tail -8 gpurun_out/r4c13_golden.log
timeout 900 python -m pytest tests/test_fullsize.py -m gpu -x -q -s -k "sd15_unet_properties or tuned_plan" > gpurun_out/r4c13_fullsize.log 2>&1; tail -6 gpurun_out/r4c13_fullsize.log
run() { python -c "import json,sys; d=json.load(open('$1')); print('$2', d['ms_per_step'], d['config']['launches_per_step'], d['config']['unet_device_ms_per_step'], d['config']['windows_ms_per_step']['each'])"; }
for i in 1 2; do
  timeout 600 python bench.py --cpu-passes 0 --windows 2 > gpurun_out/r4c13_lean_$i.json 2> gpurun_out/r4c13_lean_$i.err; run gpurun_out/r4c13_lean_$i.json "lean linears   "
  timeout 600 python bench.py --cpu-passes 0 --windows 2 --no-small-linear > gpurun_out/r4c13_gemm2_$i.json 2> gpurun_out/r4c13_gemm2_$i.err; run gpurun_out/r4c13_gemm2_$i.json "gemm2 (round 3)"
done
timeout 600 python bench.py --cpu-passes 0 --windows 0 --breakdown gpurun_out/r4c13_breakdown.txt > /dev/null 2> gpurun_out/r4c13_bd.err; head -14 gpurun_out/r4c13_breakdown.txt
