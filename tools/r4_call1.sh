#!/bin/bash
# round 4, GPU call 1: the new fused tail kernel (tests + probe), the two missing full-size parity tests, a baseline bench of this box
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_tblock_tail.py -m gpu -x -q -s > gpurun_out/r4c1_tail_tests.log 2>&1; tail -40 gpurun_out/r4c1_tail_tests.log
timeout 300 python tools/tblock_tail_probe.py > gpurun_out/r4c1_tail_probe.log 2>&1; cat gpurun_out/r4c1_tail_probe.log
timeout 900 python -m pytest tests/test_fullsize.py -m gpu -q -s -k "tuned_plan or w8a16" > gpurun_out/r4c1_parity.log 2>&1; tail -15 gpurun_out/r4c1_parity.log
timeout 600 python bench.py --breakdown gpurun_out/r4c1_breakdown.txt > gpurun_out/r4c1_bench.json 2> gpurun_out/r4c1_bench.err; cat gpurun_out/r4c1_bench.json; tail -3 gpurun_out/r4c1_bench.err
