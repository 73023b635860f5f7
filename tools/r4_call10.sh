#!/bin/bash
# round 4, GPU call 10: tail kernel v5 (GEGLU arithmetic interleaved with the ff.net.2 MFMAs) -- tests, stamps, bench A/B
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_tblock_tail.py -m gpu -x -q > gpurun_out/r4c10_tests.log 2>&1; tail -3 gpurun_out/r4c10_tests.log
for pf in 8 0; do
  echo "== prefetch workgroups $pf"; OSG_TBLOCK_PREFETCH=$pf REPS=1 SKIP_SEP=1 timeout 300 python tools/tblock_tail_probe.py 2>&1 | grep -v "^$"
done > gpurun_out/r4c10_tail_stamps.log 2>&1; cat gpurun_out/r4c10_tail_stamps.log
for i in 1 2; do
  timeout 600 python bench.py --cpu-passes 0 --windows 3 > gpurun_out/r4c10_bench_fused_$i.json 2> gpurun_out/r4c10_bench_fused_$i.err; python -c "import json,sys; d=json.load(open('gpurun_out/r4c10_bench_fused_$i.json')); print('fused', d['ms_per_step'], d['config']['launches_per_step'], d['config']['unet_device_ms_per_step'], d['config']['windows_ms_per_step']['each'])"
  timeout 600 python bench.py --cpu-passes 0 --windows 3 --no-tblock-fuse > gpurun_out/r4c10_bench_sep_$i.json 2> gpurun_out/r4c10_bench_sep_$i.err; python -c "import json,sys; d=json.load(open('gpurun_out/r4c10_bench_sep_$i.json')); print('separate', d['ms_per_step'], d['config']['launches_per_step'], d['config']['unet_device_ms_per_step'], d['config']['windows_ms_per_step']['each'])"
done
timeout 600 python bench.py --cpu-passes 0 --windows 0 --breakdown gpurun_out/r4c10_breakdown_fused.txt > /dev/null 2> gpurun_out/r4c10_bd.err; grep -n "TBlockTail\|KVPack" gpurun_out/r4c10_breakdown_fused.txt | head -8
