#!/bin/bash
# Dev tool (GPU box): A/B of the split-K finish -- reduce launch (OSG_SPLITK_TICKET=0) vs the in-kernel cooperative fold (=1) -- on the headline bench,
# each mode with its own tune table (the measured choice of the slice count depends on what a slice costs).  usage: splitk_ab.sh <tag>
mkdir -p gpurun_out
TAG=${1:-ab}
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "splitk or conv3x3 or gemm" > gpurun_out/splitk_tests_$TAG.log 2>&1; tail -3 gpurun_out/splitk_tests_$TAG.log
for mode in 0 1 0 1; do
  OSG_SPLITK_TICKET=$mode OSG_TUNE_CACHE=/tmp/osg_tune_ab_$mode.txt timeout 400 python bench.py --steps 20 --warmup 3 --cpu-passes 0 > gpurun_out/bench_splitk${mode}_$TAG.json 2> gpurun_out/bench_splitk${mode}_$TAG.err
  python - <<PY
import json
d=json.load(open("gpurun_out/bench_splitk${mode}_$TAG.json"))
print("ticket=$mode ms_per_step", d["ms_per_step"], "unet_device_ms", d["config"]["unet_device_ms_per_step"], "contraction avg us", d["roofline"]["avg_launch_us"])
PY
done
