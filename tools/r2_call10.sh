#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_golden.py tests/test_fullsize.py -m gpu -q -x 2>&1 | tail -6
for t in 1 0; do
  rm -f /tmp/tc_$t.txt
  OSG_SPLITK_TICKET=$t OSG_TUNE_CACHE=/tmp/tc_$t.txt timeout 300 python bench.py --cpu-passes 0 --profile-reps 1 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('ticket=$t ms_per_step=%.4f unet_dev=%.4f frac=%.4f' % (j['ms_per_step'], j['config']['unet_device_ms_per_step'], j['roofline']['frac']))"
done
