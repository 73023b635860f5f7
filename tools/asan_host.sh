#!/bin/bash
# Dev tool (CPU): the host library under AddressSanitizer + UBSan, driven through the no-op stub of libosgpu -- planner, providers, streaming ring
# bookkeeping, fusion passes, plan-time evaluation, the LLM flow's per-call re-plans.  Clean at the end of round 2 (75 tests + tools run) and on the final tree of round 5 (131 tests, profiles/r05_asan_host.txt).
set -e
cd "$(dirname "$0")/.."
mkdir -p /tmp/asan
g++ -std=c++20 -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fPIC -shared -Iinclude -Ionnxstream_amd/csrc/host -o /tmp/asan/libonnxstream_amd.so onnxstream_amd/csrc/host/*.cpp -ldl -lpthread
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 python -c "
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import onnxstream_amd.build as b
b.LIB_HOST = '/tmp/asan/libonnxstream_amd.so'
import pytest
sys.exit(pytest.main(['-x', '-q', '-m', 'not gpu', 'tests/test_planner_cpu.py', 'tests/test_sdpa.py', 'tests/test_llm_flow.py', 'tests/test_maskops.py', 'tests/test_shard_gloo.py', 'tests/test_golden.py', 'tests/test_movement_cpu.py', '-p', 'no:cacheprovider']))
"
