#!/bin/bash
# Dev tool (GPU box): headline bench with the measured choice timed hot (back-to-back launches) vs cold (L2/MALL evicted before every
# timed launch, OSG_TUNE_COLD=1); keeps both tune tables.
TAG=${1:-x}
mkdir -p gpurun_out
for mode in 0 1; do
  rm -f /tmp/tune_$mode.txt
  OSG_TUNE_COLD=$mode OSG_TUNE_CACHE=/tmp/tune_$mode.txt timeout 600 python bench.py --steps 40 --warmup 5 --cpu-passes 0 --profile-reps 1 --breakdown gpurun_out/breakdown_cold${mode}_$TAG.txt > gpurun_out/bench_cold${mode}_$TAG.json 2> gpurun_out/bench_cold${mode}_$TAG.err
  cp /tmp/tune_$mode.txt gpurun_out/tune_cold${mode}_$TAG.txt
  python -c "
import json;d=json.loads(open('gpurun_out/bench_cold${mode}_$TAG.json').read().strip().splitlines()[-1]);print('cold=$mode', d['ms_per_step'], d['config']['unet_device_ms_per_step'])"
done
# second replay of each table without tuning (same box, alternating) to separate box noise from choice quality
for mode in 0 1 0 1; do
  OSG_TUNE_CACHE=/tmp/tune_$mode.txt timeout 600 python bench.py --steps 40 --warmup 5 --cpu-passes 0 --profile-reps 0 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('replay table cold=$mode', d['ms_per_step'], d['config']['unet_device_ms_per_step'])"
done
