#!/bin/bash
# Dev tool (GPU box), round 6: W8A16 with resident codes through the tuned kernels.  usage: gpu_w8.sh <recipe>[,<recipe>...]   output prefix gpurun_out/r06w8
#   ktests   the W8 kernel tests + the golden W8 graph      bench   W8A16 dequantised at load vs codes resident, alternating twice, W8 shapes tuned into a copy of the shipped table
#   parity   full-size SD 1.5 W8A16 against the reference (both modes)      suite   the whole GPU suite
mkdir -p gpurun_out; export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-.}
T=gpurun_out/r06w8; export OSA_REQUIRE_ORACLE=1
pl() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); c = d["config"]
    print(sys.argv[2], "ms_per_step", d["ms_per_step"], "windows median", c["windows_ms_per_step"]["median"], "unet dev ms", c["unet_device_ms_per_step"], "launches", c.get("launches_per_step"), "misses", c["tune_table_misses"], "absmax", c["latent_absmax"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
IFS=',' read -ra RECIPES <<< "$1"; shift
for R in "${RECIPES[@]}"; do case $R in
ktests)
  timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "w8" > ${T}_kernel_tests.log 2>&1; echo "w8 kernel tests exit $?"; tail -25 ${T}_kernel_tests.log
  timeout 900 python -m pytest tests/test_golden.py -q -m gpu -k "w8" > ${T}_golden_w8.log 2>&1; echo "golden w8 exit $?"; tail -8 ${T}_golden_w8.log ;;
bench)
  export OSG_TUNE_CACHE=/tmp/tc_w8.txt; cp onnxstream_amd/tune/mi355x.txt $OSG_TUNE_CACHE
  for i in 1 2; do
    timeout 900 python bench.py --quant-weights --cpu-passes 0 --windows 2 > ${T}_bench_w8a16_$i.json 2> ${T}_bench_w8a16_$i.err; pl ${T}_bench_w8a16_$i.json "W8A16 dequantised at load"
    timeout 1500 python bench.py --quant-weights --w8-resident --cpu-passes 0 --windows 2 > ${T}_bench_w8res_$i.json 2> ${T}_bench_w8res_$i.err; pl ${T}_bench_w8res_$i.json "W8A16 codes resident"
  done
  wc -l $OSG_TUNE_CACHE; cp $OSG_TUNE_CACHE ${T}_tune_with_w8.txt
  timeout 900 python bench.py --quant-weights --w8-resident --cpu-passes 0 --windows 0 --breakdown ${T}_breakdown_w8res.txt > ${T}_bench_w8res_bd.json 2> ${T}_bench_w8res_bd.err; tail -5 ${T}_bench_w8res_bd.err ;;
libab)
  # W8-resident bench: libosgpu.so vs the library named by $1 (a file under onnxstream_amd/), alternating twice, each on its own copy of the table with W8 rows (tuned in its first run)
  ALT=$PWD/onnxstream_amd/${1:-libosgpu_w8v1.so}
  cp onnxstream_amd/tune/mi355x.txt /tmp/tc_A.txt; cp onnxstream_amd/tune/mi355x.txt /tmp/tc_B.txt
  for i in 1 2 3; do
    OSG_TUNE_CACHE=/tmp/tc_A.txt timeout 1500 python bench.py --quant-weights --w8-resident --cpu-passes 0 --windows 2 > ${T}_libab_A_$i.json 2> ${T}_libab_A_$i.err; pl ${T}_libab_A_$i.json "A libosgpu.so"
    OSG_TUNE_CACHE=/tmp/tc_B.txt OSGPU_LIB=$ALT timeout 1500 python bench.py --quant-weights --w8-resident --cpu-passes 0 --windows 2 > ${T}_libab_B_$i.json 2> ${T}_libab_B_$i.err; pl ${T}_libab_B_$i.json "B $(basename $ALT)"
  done
  OSG_TUNE_CACHE=/tmp/tc_A.txt timeout 900 python bench.py --quant-weights --cpu-passes 0 --windows 2 > ${T}_libab_w16.json 2> ${T}_libab_w16.err; pl ${T}_libab_w16.json "W8A16 dequantised at load (libosgpu.so)"
  cp /tmp/tc_A.txt ${T}_tune_with_w8.txt
  OSG_TUNE_CACHE=/tmp/tc_A.txt timeout 900 python bench.py --quant-weights --w8-resident --cpu-passes 0 --windows 0 --breakdown ${T}_breakdown_w8res.txt > ${T}_bench_w8res_bd.json 2> ${T}_bench_w8res_bd.err ;;
final)
  # evidence of the final tree: headline (default bench, W16), W8A16 at load and with codes resident on the SHIPPED table frozen (0 misses expected), alternating twice;
  # then the rocprofv3 kernel stats + in-graph timeline of the W8-resident plan
  export OSG_TUNE_CACHE=/tmp/tc_fin.txt; cp onnxstream_amd/tune/mi355x.txt $OSG_TUNE_CACHE
  timeout 900 python bench.py > ${T}_final_bench.json 2> ${T}_final_bench.err; pl ${T}_final_bench.json "headline (W16A16)"
  for i in 1 2; do
    timeout 900 python bench.py --quant-weights --frozen-table --cpu-passes 0 --windows 2 > ${T}_final_bench_w8a16_$i.json 2> ${T}_final_bench_w8a16_$i.err; pl ${T}_final_bench_w8a16_$i.json "W8A16 dequantised at load (frozen shipped table)"
    timeout 900 python bench.py --quant-weights --w8-resident --frozen-table --cpu-passes 0 --windows 2 > ${T}_final_bench_w8res_$i.json 2> ${T}_final_bench_w8res_$i.err; pl ${T}_final_bench_w8res_$i.json "W8A16 codes resident (frozen shipped table)"
  done
  rm -rf /tmp/prof_w8
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_w8 -o w8 -- python bench.py --quant-weights --w8-resident --steps 20 --warmup 2 --cpu-passes 0 --profile-reps 1 --windows 0 > ${T}_rocprof_w8res.log 2>&1
  echo "rocprofv3 exit $?"
  for f in $(find /tmp/prof_w8 -name "*kernel_stats.csv"); do cp $f ${T}_w8res_rocprofv3_kernel_stats.csv; done
  python tools/graph_trace.py $(find /tmp/prof_w8 -name "*kernel_trace.csv" | head -1) > ${T}_w8res_graph_timeline.txt 2>&1; head -16 ${T}_w8res_graph_timeline.txt ;;
pmc)
  # per-kernel counters of the W8-resident plan (tools/pmc_round4.sh with PMC_W8=1), seeded from the shipped table
  export OSG_TUNE_CACHE=/tmp/tc_pmc.txt; cp onnxstream_amd/tune/mi355x.txt $OSG_TUNE_CACHE
  PMC_W8=1 bash tools/pmc_round4.sh r06_w8res_plan 2>&1 | tail -30 ;;
lnab)  # (the "folded" leg needs the tree of branch w8-ln-fold; on main both legs run the same plan)
  # LayerNorm folded into the GEMMs on codes (osg_gemm_ln_w8) vs the standalone LayerNorm launches (--no-ln-fold: the plan of the previous commit), alternating 3x, each on its own table
  cp onnxstream_amd/tune/mi355x.txt /tmp/tc_A.txt; cp onnxstream_amd/tune/mi355x.txt /tmp/tc_B.txt
  for i in 1 2 3; do
    OSG_TUNE_CACHE=/tmp/tc_A.txt timeout 1500 python bench.py --quant-weights --w8-resident --cpu-passes 0 --windows 2 > ${T}_lnab_A_$i.json 2> ${T}_lnab_A_$i.err; pl ${T}_lnab_A_$i.json "A codes resident, LayerNorm folded"
    OSG_TUNE_CACHE=/tmp/tc_B.txt timeout 1500 python bench.py --quant-weights --w8-resident --no-ln-fold --cpu-passes 0 --windows 2 > ${T}_lnab_B_$i.json 2> ${T}_lnab_B_$i.err; pl ${T}_lnab_B_$i.json "B codes resident, LayerNorm launches"
  done
  OSG_TUNE_CACHE=/tmp/tc_A.txt timeout 900 python bench.py --quant-weights --cpu-passes 0 --windows 2 > ${T}_lnab_w16.json 2> ${T}_lnab_w16.err; pl ${T}_lnab_w16.json "W8A16 dequantised at load"
  cp /tmp/tc_A.txt ${T}_tune_with_w8ln.txt ;;
parity)
  timeout 2400 python -m pytest tests/test_fullsize.py -q -m gpu -k "w8a16" -s > ${T}_fullsize_w8.log 2>&1; echo "fullsize w8 exit $?"; grep -i "err16\|passed\|failed\|error" ${T}_fullsize_w8.log | tail -12 ;;
suite)
  timeout 3000 python -m pytest tests -q -m gpu -x > ${T}_pytest_gpu.log 2>&1; echo "gpu suite exit $?"; tail -5 ${T}_pytest_gpu.log ;;
esac; done
