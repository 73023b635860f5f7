#!/bin/bash
# round 4, GPU call 5: where does a row block's time go (stage stamps)
mkdir -p gpurun_out; export TMPDIR=/tmp
for ns in 2 3; do
  echo "== slots $ns"; OSG_TBLOCK_SLOTS=$ns REPS=1 SKIP_SEP=1 timeout 300 python tools/tblock_tail_probe.py 2>&1 | grep -v "^$"
done > gpurun_out/r4c5_tail_stamps.log 2>&1; cat gpurun_out/r4c5_tail_stamps.log
