#!/bin/bash
# Dev tool (GPU box), round 4 (round 3's tool + the identity of the tree in the file: `src_sha1`, tools/src_hash.py): per-kernel counters of THE PLAN bench.py TIMES -- run it right after `OSG_TUNE_CACHE=<table> python bench.py` in the same call:
# tools/pmc_pass.py runs with hip_autotune = 1 seeded from that table (a process that starts from a table makes the bench's choices and issues no timing
# launches), eager instead of captured (a hipGraph replay is one opaque dispatch to the counters; the kernels and their launch parameters are the same).
# Each counter set in its
# own bounded rocprofv3 pass (MI355X_MICROARCH.md: FETCH_SIZE 3 TCC slots, WRITE_SIZE 2 -- never together; --kernel-trace only, no other trace domain).
#   gpurun_out/pmc_<tag>.json : per kernel name -> {dispatches, SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE, FETCH_SIZE, WRITE_SIZE, ...} (sums)
export TMPDIR=/tmp
TAG=${1:-r4}
PASSES=${PMC_PASSES:-2}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export OSG_TUNE_CACHE=${OSG_TUNE_CACHE:-/tmp/osg_tune_cache.txt}
export PMC_AUTOTUNE=1
if [ ! -s "$OSG_TUNE_CACHE" ]; then echo "pmc_round3: no tune table at $OSG_TUNE_CACHE (run bench.py with OSG_TUNE_CACHE set first)"; exit 1; fi
timeout 300 python tools/pmc_pass.py > /tmp/pmc_prime.log 2>&1; tail -1 /tmp/pmc_prime.log
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_$i -o pmc -- python tools/pmc_pass.py > /tmp/pmc_$i.log 2>&1
  echo "pmc pass $i ($set): exit $?"; tail -2 /tmp/pmc_$i.log
done
python - $TAG $PASSES <<'PY'
import csv, glob, json, sys, collections
out = collections.defaultdict(lambda: collections.defaultdict(float))
for i in (1, 2, 3, 4):
    fs = glob.glob(f"/tmp/pmc_{i}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print("pass", i, "produced no counter file"); continue
    seen = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        out[k][r["Counter_Name"]] += float(r["Counter_Value"])
        seen[(k, r["Counter_Name"])] += 1
    for (k, c), n in seen.items():
        out[k]["dispatches"] = max(out[k]["dispatches"], n)
sys.path.insert(0, "tools"); import src_hash, os
res = {"tuned_plan": True, "w8_resident": os.environ.get("PMC_W8") == "1", "src_sha1": src_hash.src_sha1(os.getcwd()), "note": f"sums over {sys.argv[2]} eager SD1.5 batch-2 UNet passes of the TUNED plan (tools/pmc_pass.py, hip_use_graph=0, hip_autotune=1 seeded from the bench run's OSG_TUNE_CACHE table: the tile / ring / split-K choices of the timed hipGraph); FETCH_SIZE/WRITE_SIZE in KiB as rocprofv3 "
               "reports them (gfx950: FETCH_SIZE x2 for wide coalesced reads, MI355X_MICROARCH.md); SQ_VALU_MFMA_BUSY_CYCLES in cycles summed over SIMDs; "
               "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES / 32 shader engines x 256 CUs x 4 SIMDs)", "passes": int(sys.argv[2]), "kernels": {k: dict(v) for k, v in out.items()}}
json.dump(res, open(f"gpurun_out/pmc_{sys.argv[1]}.json", "w"), indent=1)
rows = []
for k, v in out.items():
    g = v.get("GRBM_GUI_ACTIVE", 0.0)
    sq = v.get("SQ_BUSY_CYCLES", 0.0) / 32.0   # (summed over the 32 shader engines)
    util = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (sq * 1024.0) if sq else 0.0
    rows.append((g, k[:70], int(v["dispatches"]), util, 2 * v.get("FETCH_SIZE", 0.0) * 1024 / 1e6, v.get("WRITE_SIZE", 0.0) * 1024 / 1e6))
for g, k, n, u, f, w in sorted(rows, reverse=True)[:24]:
    print(f"{k:70s} n={n:5d} gui_cycles={g:12.0f} mfma_util={u:6.3f} fetch(MB,x2)={f:9.1f} write(MB)={w:9.1f}")
PY
