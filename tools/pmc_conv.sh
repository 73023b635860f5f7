#!/bin/bash
# Dev tool (GPU box): SQ counters of the conv kernels on the probe shapes
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for mode in 0 1 2; do
rm -rf /tmp/pmc_$mode
PROBE_SHAPE=0 OSG_CONV3X3_BN=80 OSG_CONV3X3_SPLITS=1 OSG_CONV3X3_DBG=$mode rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmc_$mode -o pmc -- python tools/conv_probe.py > /dev/null 2>&1
f=$(find /tmp/pmc_$mode -name "*counter_collection.csv" | head -1)
python - "$f" $mode <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if "conv3x3" not in k: continue
    n = cnt[(k, "SQ_WAVE_CYCLES")]
    print("mode", sys.argv[2], k, "dispatches", n)
    wc = d["SQ_WAVE_CYCLES"] / n
    for c, v in sorted(d.items()):
        print(f"   {c:28s} {v / n:14.0f}  {100.0 * v / n / wc:6.1f}% of wave quad-cycles")
PY
done
