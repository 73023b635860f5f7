#!/bin/bash
# Dev tool (GPU box): the round-2 measurement pass -- GPU tests, smoke, bench (+ per-step breakdown), rocprofv3 kernel stats of the same command,
# the W8A8 VAE bench line.  usage: gpu_round2.sh <tag> [quick]
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
export OSG_TUNE_CACHE=/tmp/osg_tune_cache.txt   # the bench run tunes; the rocprofv3 run after it reuses the choices (no timing launches in the profile)
TAG=${1:-r2}
if [ "$2" != "quick" ]; then
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_$TAG.log 2>&1; tail -3 gpurun_out/pytest_gpu_$TAG.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1; tail -2 gpurun_out/smoke_$TAG.log
fi
timeout 900 python bench.py --breakdown gpurun_out/breakdown_$TAG.txt > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; cat gpurun_out/bench_$TAG.json; tail -3 gpurun_out/bench_$TAG.err
head -12 gpurun_out/breakdown_$TAG.txt
rm -rf /tmp/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o $TAG -- python bench.py --steps 10 --warmup 2 --cpu-passes 0 --profile-reps 1 > gpurun_out/rocprof_$TAG.log 2>&1
for f in $(find /tmp/prof_$TAG -name "*kernel_stats.csv"); do cp $f gpurun_out/rocprof_${TAG}_kernel_stats.csv; done
head -16 gpurun_out/rocprof_${TAG}_kernel_stats.csv
