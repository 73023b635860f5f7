#!/bin/bash
# round-2 GPU call 1: baseline data on the round-1 kernels
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 400 python tools/golden_table.py gpurun_out/ref16_gpubox.npz > gpurun_out/golden_table_r2a.log 2>&1; tail -25 gpurun_out/golden_table_r2a.log
timeout 300 python bench.py --cpu-passes 0 --no-autotune --breakdown gpurun_out/breakdown_r2a_noat.txt > gpurun_out/bench_r2a_noat.json 2> gpurun_out/bench_r2a_noat.err; cat gpurun_out/bench_r2a_noat.json
timeout 300 python bench.py --cpu-passes 0 --breakdown gpurun_out/breakdown_r2a.txt > gpurun_out/bench_r2a.json 2> gpurun_out/bench_r2a.err; cat gpurun_out/bench_r2a.json
timeout 700 bash tools/pmc_round.sh r2a > gpurun_out/pmc_round_r2a.log 2>&1; tail -40 gpurun_out/pmc_round_r2a.log
