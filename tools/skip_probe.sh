#!/bin/bash
# Dev tool (GPU box): marginal cost of each class of launches INSIDE the captured pass: the pass is replayed with that class left out
export TMPDIR=/tmp
mkdir -p gpurun_out
export OSG_TUNE_CACHE=/tmp/osg_tune_cache.txt
out=gpurun_out/skip_probe_${1:-r2}.txt
: > $out
for sk in "" "LayerNorm" "GroupNorm" "Concat" "Attention" "Linear" "Conv" "Linear,Conv" "Linear,Conv,Attention" "LayerNorm,GroupNorm,Concat" "Linear,Conv,Attention,LayerNorm,GroupNorm,Concat"; do
  OSG_PLAN_SKIP="$sk" timeout 200 python bench.py --mode replay --steps 60 --warmup 10 --cpu-passes 0 --profile-reps 1 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('skip=%-60s ms_per_step=%.4f' % ('$sk', j['ms_per_step']))" >> $out
done
cat $out
