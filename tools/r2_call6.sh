#!/bin/bash
# Dev tool (GPU box): headline bench without the clamp + the other builder-run lines of the round (P=4, SDXL 10-step, W8A8 VAE)
TAG=${1:-r2h}
mkdir -p gpurun_out
export OSG_TUNE_CACHE=/tmp/osg_tune_$TAG.txt
timeout 600 python bench.py --cpu-passes 0 --profile-reps 1 > gpurun_out/bench_noclamp_$TAG.json 2> gpurun_out/bench_noclamp_$TAG.err; tail -2 gpurun_out/bench_noclamp_$TAG.err; cut -c1-300 gpurun_out/bench_noclamp_$TAG.json
timeout 600 python bench.py --cpu-passes 0 --profile-reps 1 --prompts-per-gpu 4 > gpurun_out/bench_p4_$TAG.json 2> gpurun_out/bench_p4_$TAG.err; tail -2 gpurun_out/bench_p4_$TAG.err; cut -c1-300 gpurun_out/bench_p4_$TAG.json
rm -f $OSG_TUNE_CACHE
timeout 900 python bench.py --config SDXL --steps-per-image 10 --steps 20 --warmup 2 --cpu-passes 0 --profile-reps 1 > gpurun_out/bench_sdxl_$TAG.json 2> gpurun_out/bench_sdxl_$TAG.err; tail -2 gpurun_out/bench_sdxl_$TAG.err; cut -c1-300 gpurun_out/bench_sdxl_$TAG.json
