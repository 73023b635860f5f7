#!/bin/bash
# round 4, GPU call 6: counters of the tail kernel (instruction fetch? waits?)
mkdir -p gpurun_out; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --list-avail 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INSTS_[A-Z_]*\|SQ_INST_CYCLES[A-Z_]*\|SQ_WAIT[A-Z_]*\|SQ_ACTIVE_INST[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_LDS[A-Z_]*\|SQC_[A-Z_]*" | sort -u | tr '\n' ' ' > gpurun_out/r4c6_counters_avail.txt; cat gpurun_out/r4c6_counters_avail.txt; echo
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_IFETCH SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pmc_$i
  OSG_TBLOCK_SLOTS=2 NSETS=8 REPS=1 SKIP_SEP=1 timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_$i -o pmc -- python tools/tblock_tail_probe.py > /tmp/pmc_$i.log 2>&1
  echo "pmc pass $i ($set): exit $?"; tail -2 /tmp/pmc_$i.log
done
python - <<'PY' > gpurun_out/r4c6_tail_pmc.txt
import csv, glob, collections
for i in (1, 2, 3, 4):
    fs = glob.glob(f"/tmp/pmc_{i}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print("pass", i, "no counter file"); continue
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(fs[0])):
        if "tblock_tail" in r["Kernel_Name"]:
            e = acc[r["Counter_Name"]]; e[0] += float(r["Counter_Value"]); e[1] += 1
    for k, (v, n) in acc.items():
        print(f"{k:32s} per dispatch {v / n:16.1f}   ({n} dispatches)")
PY
cat gpurun_out/r4c6_tail_pmc.txt
