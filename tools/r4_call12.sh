#!/bin/bash
# round 4, GPU call 12: split-K slabs folded inside the kernel (no reduce launch) for the small-M layers only
mkdir -p gpurun_out; export TMPDIR=/tmp
run() { python -c "import json,sys; d=json.load(open('$1')); print('$2', d['ms_per_step'], d['config']['launches_per_step'], d['config']['unet_device_ms_per_step'], d['config']['windows_ms_per_step']['each'])"; }
for i in 1 2; do
  timeout 600 python bench.py --cpu-passes 0 --windows 2 > gpurun_out/r4c12_base_$i.json 2> gpurun_out/r4c12_base_$i.err; run gpurun_out/r4c12_base_$i.json "reduce launches      "
  OSG_SPLITK_TICKET=1 OSG_SPLITK_TICKET_MAXM=128 timeout 600 python bench.py --cpu-passes 0 --windows 2 > gpurun_out/r4c12_t1m128_$i.json 2> gpurun_out/r4c12_t1m128_$i.err; run gpurun_out/r4c12_t1m128_$i.json "ticket 1, M <= 128   "
  OSG_SPLITK_TICKET=2 OSG_SPLITK_TICKET_MAXM=128 timeout 600 python bench.py --cpu-passes 0 --windows 2 > gpurun_out/r4c12_t2m128_$i.json 2> gpurun_out/r4c12_t2m128_$i.err; run gpurun_out/r4c12_t2m128_$i.json "ticket 2, M <= 128   "
  OSG_SPLITK_TICKET=1 OSG_SPLITK_TICKET_MAXM=512 timeout 600 python bench.py --cpu-passes 0 --windows 2 > gpurun_out/r4c12_t1m512_$i.json 2> gpurun_out/r4c12_t1m512_$i.err; run gpurun_out/r4c12_t1m512_$i.json "ticket 1, M <= 512   "
  OSG_SPLITK_TICKET=2 OSG_SPLITK_TICKET_MAXM=512 timeout 600 python bench.py --cpu-passes 0 --windows 2 > gpurun_out/r4c12_t2m512_$i.json 2> gpurun_out/r4c12_t2m512_$i.err; run gpurun_out/r4c12_t2m512_$i.json "ticket 2, M <= 512   "
done
