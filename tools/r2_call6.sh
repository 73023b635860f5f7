#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_qu8_gpu.py -q 2>&1 | tail -5
timeout 900 python bench.py --config VAE_QU8 --steps 5 --warmup 2 --cpu-passes 0 --breakdown gpurun_out/breakdown_vae_qu8_r2b.txt > gpurun_out/bench_vae_qu8_r2b.json 2> gpurun_out/bench_vae_qu8_r2b.err; tail -3 gpurun_out/bench_vae_qu8_r2b.err; cat gpurun_out/bench_vae_qu8_r2b.json; head -14 gpurun_out/breakdown_vae_qu8_r2b.txt
