#!/bin/bash
# Dev tool (GPU box): the other BASELINE configurations, one bounded bench run each -- prompts-per-GPU batching (the reference's --num),
# W8 weights (config 3: uint8 weights dequantised to f16 tiles), SDXL-base (config 4).  Every run has its own timeout.
export TMPDIR=/tmp
export OSG_TUNE_CACHE=/tmp/osg_tune_cache.txt
TAG=${1:-r1}
mkdir -p gpurun_out
timeout 120 python bench.py --cpu-passes 0 --prompts-per-gpu 4 > gpurun_out/bench_p4_$TAG.json 2> gpurun_out/bench_p4_$TAG.err; tail -1 gpurun_out/bench_p4_$TAG.err
timeout 150 python bench.py --cpu-passes 0 --quant-weights > gpurun_out/bench_w8_$TAG.json 2> gpurun_out/bench_w8_$TAG.err; tail -1 gpurun_out/bench_w8_$TAG.err
timeout 240 python bench.py --cpu-passes 0 --config SDXL --steps-per-image 10 --steps 20 --warmup 2 > gpurun_out/bench_sdxl_$TAG.json 2> gpurun_out/bench_sdxl_$TAG.err; tail -2 gpurun_out/bench_sdxl_$TAG.err
for f in p4 w8 sdxl; do python -c "
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1], d['ms_per_step'], d['value'], d['config'].get('unet_device_ms_per_step'), d['roofline']['achieved'])
except Exception as e: print(sys.argv[1], 'FAILED', e)" gpurun_out/bench_${f}_$TAG.json; done
