#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== table every call"; timeout 300 python tools/gnconv_probe.py 2>&1 | tail -9
echo "== table once (conv with the affine in its loaders, kernel only)"; OSG_GNCONV_TABLE_ONCE=1 timeout 300 python tools/gnconv_probe.py 2>&1 | tail -9
