#!/bin/bash
# Dev tool (GPU box): where do the wave-cycles of the self-attention kernel go?  One SQ counter set, kernel-trace only.
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf /tmp/pmc_attn
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d /tmp/pmc_attn -o a -- python tools/attn_probe.py > /tmp/pmc_attn.log 2>&1
tail -3 /tmp/pmc_attn.log
rm -rf /tmp/pmc_attn2
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_attn2 -o a -- python tools/attn_probe.py > /tmp/pmc_attn2.log 2>&1
tail -2 /tmp/pmc_attn2.log
python - <<'PY'
import csv, glob, collections
for d in ("/tmp/pmc_attn", "/tmp/pmc_attn2"):
    out = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "attn" not in r["Kernel_Name"] or "kernel" not in r["Kernel_Name"]: continue
            k = r["Kernel_Name"].split("(")[0].split("::")[-1][-44:] + " grid " + r.get("Grid_Size", "?")
            out[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k, v in out.items():
        disp = max(c for (kk, _), c in n.items() if kk == k)
        print(k, "dispatches", disp, {c: round(x / disp) for c, x in v.items()})
PY
