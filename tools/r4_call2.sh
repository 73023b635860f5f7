#!/bin/bash
# round 4, GPU call 2: the fused tail in the planner -- kernel probe (device time), golden chains, full-size parity of the default plan, bench A/B
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/tblock_tail_probe.py > gpurun_out/r4c2_tail_probe.log 2>&1; cat gpurun_out/r4c2_tail_probe.log
timeout 600 python -m pytest tests/test_tblock_tail.py tests/test_golden.py -m gpu -x -q -s -k "tblock or chains or transformer_block" > gpurun_out/r4c2_tests.log 2>&1; grep -v "^x1\|^ln\|^q \|^a2\|^x2\|^x3\|^y " gpurun_out/r4c2_tests.log | tail -30
timeout 900 python -m pytest tests/test_fullsize.py -m gpu -x -q -s -k "sd15_unet_properties or tuned_plan" > gpurun_out/r4c2_fullsize.log 2>&1; tail -8 gpurun_out/r4c2_fullsize.log
for i in 1 2; do
  timeout 600 python bench.py --cpu-passes 0 --windows 3 > gpurun_out/r4c2_bench_fused_$i.json 2> gpurun_out/r4c2_bench_fused_$i.err; python -c "import json,sys; d=json.load(open('gpurun_out/r4c2_bench_fused_$i.json')); print('fused', d['ms_per_step'], d['config']['launches_per_step'], d['config']['unet_device_ms_per_step'], d['config']['windows_ms_per_step']['each'])"
  timeout 600 python bench.py --cpu-passes 0 --windows 3 --no-tblock-fuse > gpurun_out/r4c2_bench_sep_$i.json 2> gpurun_out/r4c2_bench_sep_$i.err; python -c "import json,sys; d=json.load(open('gpurun_out/r4c2_bench_sep_$i.json')); print('separate', d['ms_per_step'], d['config']['launches_per_step'], d['config']['unet_device_ms_per_step'], d['config']['windows_ms_per_step']['each'])"
done
timeout 600 python bench.py --cpu-passes 0 --windows 0 --breakdown gpurun_out/r4c2_breakdown_fused.txt > /dev/null 2> gpurun_out/r4c2_bd.err; grep -n "TBlockTail\|KVPack" gpurun_out/r4c2_breakdown_fused.txt | head; head -12 gpurun_out/r4c2_breakdown_fused.txt
