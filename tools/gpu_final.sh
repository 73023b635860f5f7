#!/bin/bash
# Dev tool (GPU box): end-of-round validation -- the whole GPU suite, smoke, the headline bench (+ breakdown), the same bench with a ONE-rank RCCL
# process group (OSA_BENCH_FORCE_DIST=1: every collective of the N > 1 flow on the real backend), the W8A8 VAE line.   usage: gpu_final.sh <tag>
mkdir -p gpurun_out
export TMPDIR=/tmp
export OSG_TUNE_CACHE=/tmp/osg_tune_cache.txt
TAG=${1:-final}
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_$TAG.log 2>&1; tail -3 gpurun_out/pytest_gpu_$TAG.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1; tail -2 gpurun_out/smoke_$TAG.log
timeout 600 python bench.py --breakdown gpurun_out/breakdown_$TAG.txt > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; cat gpurun_out/bench_$TAG.json; tail -2 gpurun_out/bench_$TAG.err
OSA_BENCH_FORCE_DIST=1 timeout 400 python bench.py --steps 10 --warmup 2 --cpu-passes 0 > gpurun_out/bench_rccl1_$TAG.json 2> gpurun_out/bench_rccl1_$TAG.err; echo "rccl one-rank rc=$?"; cut -c1-300 gpurun_out/bench_rccl1_$TAG.json; tail -3 gpurun_out/bench_rccl1_$TAG.err
timeout 200 python bench.py --config VAE_QU8 --steps 20 --warmup 3 > gpurun_out/bench_vae_qu8_$TAG.json 2>/dev/null; cut -c1-260 gpurun_out/bench_vae_qu8_$TAG.json
