#!/bin/bash
TAG=${1:-r2f}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "group_norm" > gpurun_out/pytest_gn_$TAG.log 2>&1; tail -5 gpurun_out/pytest_gn_$TAG.log
for off in 1 0 1 0; do
  if [ $off = 1 ]; then export OSG_GN_CLUSTER_OFF=1; else unset OSG_GN_CLUSTER_OFF; fi
  timeout 600 python bench.py --steps 40 --warmup 5 --cpu-passes 0 --profile-reps 1 --no-autotune --breakdown gpurun_out/breakdown_gncl${off}_$TAG.txt 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('cluster_off=$off', d['ms_per_step'], d['config']['unet_device_ms_per_step'], d['config']['launches_per_step'])"
  grep GroupNorm gpurun_out/breakdown_gncl${off}_$TAG.txt | head -1
done
