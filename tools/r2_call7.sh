#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
export OSG_TUNE_CACHE=/tmp/osg_tune_cache.txt
for fl in "" "--ln-fold" "--no-autotune" "--no-autotune --ln-fold"; do
  timeout 300 python bench.py --cpu-passes 0 --profile-reps 1 $fl 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('flags=%-28s ms_per_step=%.4f unet_dev=%.4f launches=%d' % ('$fl', j['ms_per_step'], j['config']['unet_device_ms_per_step'], j['config']['launches_per_step']))"
done
