#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
run() { python -c "import json,sys; d=json.load(open('$1')); print('$2', d['ms_per_step'], d['config']['launches_per_step'], d['config']['unet_device_ms_per_step'], d['config']['windows_ms_per_step']['each'])"; }
for i in 1 2 3; do
  timeout 600 python bench.py --cpu-passes 0 --windows 2 --small-linear 1 > gpurun_out/r4c14_l1_$i.json 2> gpurun_out/r4c14_l1_$i.err; run gpurun_out/r4c14_l1_$i.json "lean where faster (1)"
  timeout 600 python bench.py --cpu-passes 0 --windows 2 --small-linear 0 > gpurun_out/r4c14_l0_$i.json 2> gpurun_out/r4c14_l0_$i.err; run gpurun_out/r4c14_l0_$i.json "gemm2 only        (0)"
done
