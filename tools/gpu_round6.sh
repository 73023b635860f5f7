#!/bin/bash
# Dev tool (GPU box), round 6: parameterised runner for this round's gpurun calls (rounds 4-5: gpu_round.sh, whose recipes it reuses through `old <recipes> [args]`).
# usage: gpu_round6.sh <recipe>[,<recipe>...] [args]     output prefix gpurun_out/r06
#   floor4      tools/floor_probe4 (built here, travels): what a trivial node costs cold / hot / with its code prefetched
#   ref16       the reference's fp16 / fp32 outputs of the full-size nets on THIS host's CPU -> gpurun_out/ref16_fullsize_epyc.npz
#   libab <lib> / libs <lib> ...   whole builds of libosgpu.so (files under onnxstream_amd/, OSGPU_LIB) alternating on the headline bench
#   pipeprobe / retune / extratrivial   see the recipes
#   old ...     tools/gpu_round.sh with ROUND=r06
mkdir -p gpurun_out; export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-.}
export ROUND=r06; T=gpurun_out/r06; export OSA_REQUIRE_ORACLE=1
IFS=',' read -ra RECIPES <<< "$1"; shift
for R in "${RECIPES[@]}"; do case $R in
floor4)
  timeout 300 tools/_build/floor_probe4 9 > ${T}_floor_probe4.txt 2>&1; echo "floor_probe4 exit $?"; cat ${T}_floor_probe4.txt ;;
ref16)
  timeout 1500 python tools/ref16_fullsize.py epyc gpurun_out ${1:-sd15,sd15_w8,sdxl,vae} > ${T}_ref16_epyc.log 2>&1; echo "ref16 exit $?"; tail -8 ${T}_ref16_epyc.log ;;
pipeprobe)
  # the k loop pipelined across its barrier (libosgpu_pipe.so = the library built with -DOSG_GEMM_PIPE=1): kernel tests on it, then the k-loop probe of both builds, alternating
  OSGPU_LIB=$PWD/onnxstream_amd/libosgpu_pipe.so timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm or conv or 160 or two_wave or splitk" > ${T}_pipe_tests.log 2>&1; echo "pipe kernel tests exit $?"; tail -3 ${T}_pipe_tests.log
  for rep in 1 2; do
    timeout 600 python tools/gemm_kloop_probe.py > ${T}_kloop_base_$rep.txt 2>&1
    OSGPU_LIB=$PWD/onnxstream_amd/libosgpu_pipe.so timeout 600 python tools/gemm_kloop_probe.py > ${T}_kloop_pipe_$rep.txt 2>&1
  done; tail -3 ${T}_kloop_pipe_2.txt ;;
libab)
  # headline bench, alternating: libosgpu.so vs the library named by $1 (default libosgpu_pipe.so); first both on the shipped tune table, then each on a table it tunes itself
  ALT=$PWD/onnxstream_amd/${1:-libosgpu_pipe.so}
  pl() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); c = d["config"]
    print(sys.argv[2], "ms_per_step", d["ms_per_step"], "windows median", c["windows_ms_per_step"]["median"], "unet dev ms", c["unet_device_ms_per_step"], "misses", c["tune_table_misses"], "absmax", c["latent_absmax"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  }
  for i in 1 2; do
    export OSG_TUNE_CACHE=/tmp/tc_a.txt; cp onnxstream_amd/tune/mi355x.txt $OSG_TUNE_CACHE
    timeout 600 python bench.py --cpu-passes 0 --windows 2 > ${T}_libab_A_$i.json 2> ${T}_libab_A_$i.err; pl ${T}_libab_A_$i.json "A libosgpu.so shipped table"
    export OSG_TUNE_CACHE=/tmp/tc_b.txt; cp onnxstream_amd/tune/mi355x.txt $OSG_TUNE_CACHE
    OSGPU_LIB=$ALT timeout 600 python bench.py --cpu-passes 0 --windows 2 > ${T}_libab_B_$i.json 2> ${T}_libab_B_$i.err; pl ${T}_libab_B_$i.json "B $(basename $ALT) shipped table"
  done
  export OSG_TUNE_CACHE=/tmp/tc_a2.txt; rm -f $OSG_TUNE_CACHE
  timeout 900 python bench.py --cpu-passes 0 --windows 2 > ${T}_libab_A_tuned.json 2> ${T}_libab_A_tuned.err; pl ${T}_libab_A_tuned.json "A libosgpu.so own table"
  export OSG_TUNE_CACHE=/tmp/tc_b2.txt; rm -f $OSG_TUNE_CACHE
  OSGPU_LIB=$ALT timeout 900 python bench.py --cpu-passes 0 --windows 2 > ${T}_libab_B_tuned.json 2> ${T}_libab_B_tuned.err; pl ${T}_libab_B_tuned.json "B $(basename $ALT) own table"
  cp /tmp/tc_b2.txt ${T}_tune_alt.txt; cp /tmp/tc_a2.txt ${T}_tune_base.txt
  export OSG_TUNE_CACHE=/tmp/tc_a2.txt; timeout 600 python bench.py --cpu-passes 0 --windows 2 > ${T}_libab_A_tuned2.json 2> ${T}_libab_A_tuned2.err; pl ${T}_libab_A_tuned2.json "A libosgpu.so own table (again)"
  export OSG_TUNE_CACHE=/tmp/tc_b2.txt; OSGPU_LIB=$ALT timeout 600 python bench.py --cpu-passes 0 --windows 2 > ${T}_libab_B_tuned2.json 2> ${T}_libab_B_tuned2.err; pl ${T}_libab_B_tuned2.json "B $(basename $ALT) own table (again)"
  ;;
libs)
  # headline bench on the shipped tune table, the libraries named in $@ (files under onnxstream_amd/) alternating, 2 rounds
  pl2() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); c = d["config"]
    print(sys.argv[2], "ms_per_step", d["ms_per_step"], "windows median", c["windows_ms_per_step"]["median"], "unet dev ms", c["unet_device_ms_per_step"], "misses", c["tune_table_misses"], "absmax", c["latent_absmax"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  }
  for i in 1 2; do for L in "$@"; do
    export OSG_TUNE_CACHE=/tmp/tc_$L.txt; cp onnxstream_amd/tune/mi355x.txt $OSG_TUNE_CACHE
    OSGPU_LIB=$PWD/onnxstream_amd/$L timeout 600 python bench.py --cpu-passes 0 --windows 2 > ${T}_libs_${L}_$i.json 2> ${T}_libs_${L}_$i.err; pl2 ${T}_libs_${L}_$i.json "$L"
  done; done; break ;;
retune)
  # the shipped tune table re-measured on this tree: empty table -> headline (its rows first), again (0 misses), then SDXL and 4 prompts per GPU append theirs -> ${T}_tune_extended.txt
  export OSG_TUNE_CACHE=/tmp/tc_final.txt; rm -f $OSG_TUNE_CACHE
  timeout 900 python bench.py --cpu-passes 0 --windows 2 > ${T}_retune_headline.json 2> ${T}_retune_headline.err; wc -l $OSG_TUNE_CACHE; cp $OSG_TUNE_CACHE ${T}_tune_headline.txt
  timeout 900 python bench.py --cpu-passes 0 --windows 2 > ${T}_retune_headline2.json 2> ${T}_retune_headline2.err
  timeout 900 python bench.py --config SDXL --steps-per-image 10 --steps 20 --warmup 2 --cpu-passes 0 --windows 0 > ${T}_retune_sdxl.json 2> ${T}_retune_sdxl.err; wc -l $OSG_TUNE_CACHE
  timeout 900 python bench.py --prompts-per-gpu 4 --cpu-passes 0 --windows 0 > ${T}_retune_p4.json 2> ${T}_retune_p4.err; wc -l $OSG_TUNE_CACHE
  cp $OSG_TUNE_CACHE ${T}_tune_extended.txt ;;
extratrivial)
  # what a trivial node costs INSIDE the captured pass: k extra trivial launches behind every plan step (OSG_PROBE_EXTRA_TRIVIAL, plan_run.cpp), unprofiled, alternating twice
  export OSG_TUNE_CACHE=/tmp/tc_x.txt; cp onnxstream_amd/tune/mi355x.txt $OSG_TUNE_CACHE
  for i in 1 2; do for k in 0 1 2 4; do
    OSG_PROBE_EXTRA_TRIVIAL=$k timeout 600 python bench.py --cpu-passes 0 --windows 0 --steps 20 --warmup 3 > ${T}_extra_${k}_$i.json 2> ${T}_extra_${k}_$i.err
    python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); print('extra trivial launches per step', sys.argv[2], 'ms_per_step', d['ms_per_step'], 'unet dev ms', d['config']['unet_device_ms_per_step'])" ${T}_extra_${k}_$i.json $k
  done; done ;;
old)
  bash tools/gpu_round.sh "$@"; break ;;
*) echo "unknown recipe $R" ;;
esac; done
