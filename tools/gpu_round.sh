#!/bin/bash
# Dev tool (GPU box), rounds 4-5: ONE parameterised runner instead of a one-shot script per gpurun call.  usage: gpu_round.sh <recipe> [args]; several recipes: "a,b,c"
# (output prefix gpurun_out/$ROUND, default r05)
#   tests            the whole GPU suite + smoke()
#   bench            the default bench line + per-launch breakdown           -> gpurun_out/r04_bench.json, r04_breakdown.txt
#   ab "<A>" "<B>"   bench.py with flag set A vs flag set B, alternating 2x   (e.g. ab "" "--no-tblock-fuse")
#   configs          the other BASELINE configs: SDXL, 4 prompts, W8A16, W8A16 resident codes, W8A8 VAE
#   tworank          the N = 2 flow on one GPU (OSA_BENCH_ONE_GPU=1, gloo), as the driver launches it (no OSG_TUNE_CACHE in the environment): line format, frozen shipped tune table on both ranks
#   prof             rocprofv3 --kernel-trace --stats of the bench command + the in-graph timeline (tools/graph_trace.py)
#   pmc              per-kernel counters of the timed plan, each set in its own --pmc pass (tools/pmc_round4.sh)
#   tailprobe        osg_tblock_tail: per-launch time (cold / hot weights) + stage stamps for 64- / 32-row blocks, 1 / 2 weight tiles ahead, with / without prefetching workgroups
#   tailtests        the tail kernel's tests + the golden chains
#   abenv "<env A>" "<env B>" ...  bench.py under each environment (X=1 Y=2 strings; use a dummy variable for 'default'), alternating 2x
mkdir -p gpurun_out; export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-.}
T=gpurun_out/${ROUND:-r05}; export OSA_REQUIRE_ORACLE=1
line() { python -c "import json,sys; d=json.load(open('$1')); c=d['config']; print('$2', 'ms_per_step', d['ms_per_step'], 'value', d['value'], d['unit'], 'launches', c.get('launches_per_step'), 'unet_device_ms', c.get('unet_device_ms_per_step'), 'windows', (c.get('windows_ms_per_step') or {}).get('each'), 'frac', (d.get('roofline') or {}).get('frac'))"; }
IFS=',' read -ra RECIPES <<< "$1"; shift; ARGS=("$@")
for R in "${RECIPES[@]}"; do case $R in
tests)
  timeout 1200 python -m pytest tests -m gpu -x -q > ${T}_pytest_gpu.log 2>&1; tail -3 ${T}_pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > ${T}_smoke.log 2>&1; tail -2 ${T}_smoke.log ;;
bench)
  export OSG_TUNE_CACHE=/tmp/osg_tune_cache.txt; rm -f $OSG_TUNE_CACHE; cp onnxstream_amd/tune/mi355x.txt $OSG_TUNE_CACHE
  timeout 900 python bench.py --breakdown ${T}_breakdown.txt > ${T}_bench.json 2> ${T}_bench.err; cat ${T}_bench.json; tail -2 ${T}_bench.err; head -16 ${T}_breakdown.txt ;;
ab)
  for i in 1 2; do
    timeout 600 python bench.py --cpu-passes 0 --windows 2 ${ARGS[0]} > ${T}_ab_A_$i.json 2> ${T}_ab_A_$i.err; line ${T}_ab_A_$i.json "A [${ARGS[0]}]"
    timeout 600 python bench.py --cpu-passes 0 --windows 2 ${ARGS[1]} > ${T}_ab_B_$i.json 2> ${T}_ab_B_$i.err; line ${T}_ab_B_$i.json "B [${ARGS[1]}]"
  done ;;
configs)
  timeout 900 python bench.py --config SDXL --steps-per-image 10 --steps 20 --warmup 2 --cpu-passes 0 > ${T}_bench_sdxl.json 2> ${T}_bench_sdxl.err; line ${T}_bench_sdxl.json "SDXL 1024x1024 10-step"
  timeout 600 python bench.py --prompts-per-gpu 4 --cpu-passes 0 > ${T}_bench_p4.json 2> ${T}_bench_p4.err; line ${T}_bench_p4.json "4 prompts per GPU"
  timeout 600 python bench.py --quant-weights --cpu-passes 0 > ${T}_bench_w8a16.json 2> ${T}_bench_w8a16.err; line ${T}_bench_w8a16.json "W8A16 (dequantised at load)"
  timeout 600 python bench.py --quant-weights --w8-resident --cpu-passes 0 > ${T}_bench_w8res.json 2> ${T}_bench_w8res.err; line ${T}_bench_w8res.json "W8A16 resident codes"
  timeout 300 python bench.py --config VAE_QU8 --steps 20 --warmup 3 > ${T}_bench_vae_qu8.json 2> ${T}_bench_vae_qu8.err; cut -c1-300 ${T}_bench_vae_qu8.json ;;
tworank)
  OSA_BENCH_ONE_GPU=1 timeout 900 env -u OSG_TUNE_CACHE -u OSG_TUNE_FROZEN python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 20 --warmup 2 --cpu-passes 0 --profile-reps 0 > ${T}_two_rank.json 2> ${T}_two_rank.err
  echo "two ranks on one GPU: exit $?"; tail -3 ${T}_two_rank.err; cut -c1-700 ${T}_two_rank.json ;;
prof)
  export OSG_TUNE_CACHE=/tmp/osg_tune_cache.txt; [ -s $OSG_TUNE_CACHE ] || cp onnxstream_amd/tune/mi355x.txt $OSG_TUNE_CACHE
  rm -rf /tmp/prof_f
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f -o fin -- python bench.py --steps 20 --warmup 2 --cpu-passes 0 --profile-reps 1 --windows 0 > ${T}_rocprof.log 2>&1
  echo "rocprofv3 exit $?"; grep -c "Memory access fault" ${T}_rocprof.log
  for f in $(find /tmp/prof_f -name "*kernel_stats.csv"); do cp $f ${T}_rocprofv3_kernel_stats.csv; done
  python tools/graph_trace.py $(find /tmp/prof_f -name "*kernel_trace.csv" | head -1) > ${T}_graph_timeline.txt 2>&1; head -24 ${T}_graph_timeline.txt ;;
pmc)
  export OSG_TUNE_CACHE=/tmp/osg_tune_cache.txt; [ -s $OSG_TUNE_CACHE ] || cp onnxstream_amd/tune/mi355x.txt $OSG_TUNE_CACHE
  bash tools/pmc_round4.sh ${ROUND:-r05}_tuned_plan 2>&1 | tail -30 ;;
tailprobe)
  for cfg in "64 2 8" "32 2 8" "32 2 0" "32 3 8" "32 3 0" "64 2 0"; do set -- $cfg; echo "== rows per block $1, weight tiles in flight $(($2 - 1)), prefetching workgroups $3"
    ROWS=$1 OSG_TBLOCK_NS=$2 OSG_TBLOCK_PREFETCH=$3 REPS=2 timeout 300 python tools/tblock_tail_probe.py 2>&1 | grep -v "^$"; done > ${T}_tail_probe.log 2>&1; cat ${T}_tail_probe.log ;;
kerneltests)   # kerneltests "<pytest -k expression>": part of tests/test_gpu_kernels.py
  timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "${ARGS[0]}" > ${T}_kernel_tests.log 2>&1; tail -4 ${T}_kernel_tests.log ;;
tailtests)
  timeout 900 python -m pytest tests/test_tblock_tail.py tests/test_golden.py -m gpu -x -q -k "tblock or chains" > ${T}_tail_tests.log 2>&1; tail -4 ${T}_tail_tests.log ;;
p4rows)   # 4 prompts per GPU (M = 32 768 rows at the 64x64 level): 64- vs 32-row blocks in the tail kernel
  for i in 1 2; do for r in 64 32; do
    OSG_TBLOCK_ROWS=$r timeout 600 python bench.py --prompts-per-gpu 4 --cpu-passes 0 --windows 2 > ${T}_p4rows_${r}_$i.json 2> ${T}_p4rows_${r}_$i.err; line ${T}_p4rows_${r}_$i.json "[4 prompts, $r-row blocks]"; done; done ;;
abenv)   # abenv "<env A>" "<env B>" [more env sets ...]: bench.py under each environment, alternating 2x
  for i in 1 2; do n=0; for e in "${ARGS[@]}"; do n=$((n + 1))
    env $e timeout 600 python bench.py --cpu-passes 0 --windows 2 > ${T}_abenv_${n}_$i.json 2> ${T}_abenv_${n}_$i.err; line ${T}_abenv_${n}_$i.json "[$e]"; done; done ;;
attn)   # attention: kernel tests + golden chains, then the probe with the round-2 kernel (OSG_ATTN_V1=1) and the round-5 kernel
  timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_sdpa.py tests/test_golden.py -m gpu -x -q -k "attention or sdpa or chains" > ${T}_attn_tests.log 2>&1; tail -5 ${T}_attn_tests.log
  for v in "OSG_ATTN_V1=1" "OSG_ATTN_V1=0" ${ARGS[@]}; do env $v timeout 300 python tools/attn_probe.py 2>&1 | sed "s/^/[$v] /"; done > ${T}_attn_probe.txt; cat ${T}_attn_probe.txt ;;
abenv1)   # like abenv, one round only
  n=0; for e in "${ARGS[@]}"; do n=$((n + 1))
    env $e timeout 600 python bench.py --cpu-passes 0 --windows 2 > ${T}_abenv1_${n}.json 2> ${T}_abenv1_${n}.err; line ${T}_abenv1_${n}.json "[$e]"; done ;;
ktests)   # the kernel-level GPU tests
  timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_tblock_tail.py tests/test_sdpa.py -m gpu -x -q > ${T}_kernel_tests.log 2>&1; tail -6 ${T}_kernel_tests.log ;;
gtable)   # error table of every golden case under the deterministic plans + the reference's fp16 output on this host
  timeout 900 python tools/golden_table.py ${T}_ref16_host.npz > ${T}_golden_table.txt 2> ${T}_golden_table.err; tail -8 ${T}_golden_table.txt ;;
retune)   # retune "<tag>" "<env>": the headline bench from an EMPTY tune table under <env>; the table it measured -> gpurun_out (then SDXL / 4 prompts appended by `extend`)
  tag=${ARGS[0]}; rm -f /tmp/osg_tune_$tag.txt
  env ${ARGS[1]} OSG_TUNE_CACHE=/tmp/osg_tune_$tag.txt timeout 900 python bench.py --cpu-passes 0 --windows 3 > ${T}_retune_$tag.json 2> ${T}_retune_$tag.err; line ${T}_retune_$tag.json "[retune $tag: ${ARGS[1]}]"
  env ${ARGS[1]} OSG_TUNE_CACHE=/tmp/osg_tune_$tag.txt timeout 900 python bench.py --cpu-passes 0 --windows 3 > ${T}_retune_${tag}_2.json 2> ${T}_retune_${tag}_2.err; line ${T}_retune_${tag}_2.json "[seeded from it: $tag]"
  cp /tmp/osg_tune_$tag.txt ${T}_tune_$tag.txt; wc -l ${T}_tune_$tag.txt; ARGS=("${ARGS[@]:2}") ;;
extend)   # extend "<tag>": SDXL, the W8A16 / 4-prompt variants run on the table of `retune <tag>`, appending the shapes they add
  tag=${ARGS[0]}; export OSG_TUNE_CACHE=/tmp/osg_tune_$tag.txt
  timeout 900 python bench.py --config SDXL --steps-per-image 10 --steps 20 --warmup 2 --cpu-passes 0 --windows 2 > ${T}_bench_sdxl.json 2> ${T}_bench_sdxl.err; line ${T}_bench_sdxl.json "SDXL 1024x1024 10-step"
  timeout 600 python bench.py --prompts-per-gpu 4 --cpu-passes 0 --windows 2 > ${T}_bench_p4.json 2> ${T}_bench_p4.err; line ${T}_bench_p4.json "4 prompts per GPU"
  cp /tmp/osg_tune_$tag.txt ${T}_tune_${tag}_extended.txt; wc -l ${T}_tune_${tag}_extended.txt; unset OSG_TUNE_CACHE; ARGS=("${ARGS[@]:1}") ;;
foldcheck)   # the in-kernel split-K fold: its kernel tests (bounded), then -- only if they pass -- the headline bench retuned with the fold among the candidates
  timeout 240 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "splitk_fold or output_views or statistics_from" > ${T}_fold_tests.log 2>&1; rc=$?; tail -4 ${T}_fold_tests.log
  if [ $rc -eq 0 ]; then
    rm -f /tmp/osg_tune_fold.txt
    OSG_SPLITK_FOLD=1 OSG_TUNE_CACHE=/tmp/osg_tune_fold.txt timeout 300 python bench.py --cpu-passes 0 --windows 3 > ${T}_retune_fold.json 2> ${T}_retune_fold.err; line ${T}_retune_fold.json "[retune fold]"
    OSG_SPLITK_FOLD=1 OSG_TUNE_CACHE=/tmp/osg_tune_fold.txt timeout 200 python bench.py --cpu-passes 0 --windows 3 > ${T}_retune_fold_2.json 2> ${T}_retune_fold_2.err; line ${T}_retune_fold_2.json "[seeded from it: fold]"
    cp /tmp/osg_tune_fold.txt ${T}_tune_fold.txt; wc -l ${T}_tune_fold.txt; grep -c " 1[0-9] [0-9]* [0-9]* [0-9]* [0-9.]*$\| 2[0-9] [0-9]* [0-9]* [0-9]* [0-9.]*$" ${T}_tune_fold.txt
  else echo "fold tests failed: no retune"; fi ;;
paritytable)   # every test that uses the triangulated rule, without -x: the table of margins (tests/parity.py prints one line per case)
  timeout 600 python -m pytest tests/test_golden.py tests/test_fullsize.py tests/test_real_graph.py -m gpu -q -s -k "whole_nets or chains or full_size or yolov8 or streamed or w8 or measured or sd15" > ${T}_parity_table.log 2>&1
  grep "leg (\|passed\|failed\|FAILED\|Error" ${T}_parity_table.log | cut -c1-260 ;;
foldab)   # the headline bench retuned with the in-kernel split-K fold among the candidates, then alternating 2x against the table measured WITHOUT it (gpurun_out/r05_tune_nofold.txt)
  rm -f /tmp/osg_tune_fold.txt
  OSG_SPLITK_FOLD=1 OSG_TUNE_CACHE=/tmp/osg_tune_fold.txt timeout 300 python bench.py --cpu-passes 0 --windows 3 > ${T}_retune_fold.json 2> ${T}_retune_fold.err; line ${T}_retune_fold.json "[retune fold]"
  cp /tmp/osg_tune_fold.txt ${T}_tune_fold.txt; wc -l ${T}_tune_fold.txt; echo "rows with the fold bit:"; awk '{ if (int($15 / 16) % 2 == 1) n++ } END { print n + 0 }' ${T}_tune_fold.txt
  cp tools/tune_nofold_r05.txt /tmp/osg_tune_nofold.txt
  for i in 1 2; do
    OSG_SPLITK_FOLD=0 OSG_TUNE_CACHE=/tmp/osg_tune_nofold.txt timeout 200 python bench.py --cpu-passes 0 --windows 3 > ${T}_foldab_A_$i.json 2> ${T}_foldab_A_$i.err; line ${T}_foldab_A_$i.json "A [reduce launches]"
    OSG_SPLITK_FOLD=1 OSG_TUNE_CACHE=/tmp/osg_tune_fold.txt timeout 200 python bench.py --cpu-passes 0 --windows 3 > ${T}_foldab_B_$i.json 2> ${T}_foldab_B_$i.err; line ${T}_foldab_B_$i.json "B [fold among the candidates]"
  done ;;
gnprof)   # GroupNorm per instantiation: the in-graph timeline with the statistics coming from the producing convolutions (hip_gn_stats 1) -- the default plan's is `prof`
  export OSG_TUNE_CACHE=/tmp/osg_tune_cache_gn.txt; cp onnxstream_amd/tune/mi355x.txt $OSG_TUNE_CACHE
  rm -rf /tmp/prof_gn
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_gn -o gn -- python bench.py --gn-stats 1 --steps 20 --warmup 2 --cpu-passes 0 --profile-reps 1 --windows 0 > ${T}_rocprof_gn.log 2>&1
  echo "rocprofv3 (gn stats on) exit $?"
  for f in $(find /tmp/prof_gn -name "*kernel_stats.csv"); do cp $f ${T}_gn_stats_on_kernel_stats.csv; done
  python tools/graph_trace.py $(find /tmp/prof_gn -name "*kernel_trace.csv" | head -1) > ${T}_gn_stats_on_graph_timeline.txt 2>&1; head -26 ${T}_gn_stats_on_graph_timeline.txt; unset OSG_TUNE_CACHE ;;
r4vs5)   # the round-4 kernels' equivalent (attention v1, reduce launches only, the table tuned without the fold) against this tree's default, alternating 2x on one box; then SDXL with attention v1 / v2
  cp tools/tune_nofold_r05.txt /tmp/osg_tune_nofold.txt; cp onnxstream_amd/tune/mi355x.txt /tmp/osg_tune_fold.txt
  for i in 1 2; do
    OSG_ATTN_V1=1 OSG_SPLITK_FOLD=0 OSG_TUNE_CACHE=/tmp/osg_tune_nofold.txt timeout 200 python bench.py --cpu-passes 0 --windows 3 > ${T}_r4vs5_A_$i.json 2> ${T}_r4vs5_A_$i.err; line ${T}_r4vs5_A_$i.json "A [attention v1, reduce launches]"
    OSG_TUNE_CACHE=/tmp/osg_tune_fold.txt timeout 200 python bench.py --cpu-passes 0 --windows 3 > ${T}_r4vs5_B_$i.json 2> ${T}_r4vs5_B_$i.err; line ${T}_r4vs5_B_$i.json "B [round 5 default]"
  done
  rm -f /tmp/osg_tune_sdxl.txt; cp onnxstream_amd/tune/mi355x.txt /tmp/osg_tune_sdxl.txt
  for v in 0 1 0 1; do
    OSG_ATTN_V1=$v OSG_TUNE_CACHE=/tmp/osg_tune_sdxl.txt timeout 400 python bench.py --config SDXL --steps-per-image 10 --steps 20 --warmup 2 --cpu-passes 0 --windows 2 > ${T}_sdxl_attn_v1_$v.json 2> ${T}_sdxl_attn_v1_$v.err; line ${T}_sdxl_attn_v1_$v.json "SDXL [OSG_ATTN_V1=$v]"
  done ;;
frozen)   # N = 1 as the ranks of an N > 1 job run (shipped table, OSG_TUNE_FROZEN=1): the same plan, the same time, 0 misses
  timeout 400 python bench.py --frozen-table --cpu-passes 0 --windows 3 > ${T}_bench_frozen.json 2> ${T}_bench_frozen.err; line ${T}_bench_frozen.json "[--frozen-table]"
  python -c "import json; c=json.load(open('${T}_bench_frozen.json'))['config']; print('tune_table_frozen', c['tune_table_frozen'], 'tune_table_misses', c['tune_table_misses'])" ;;
sdxlprof)   # SDXL 1024x1024: the in-graph timeline of one replayed pass (rocprofv3 --kernel-trace of the bench command)
  export OSG_TUNE_CACHE=/tmp/osg_tune_cache_sdxl.txt; [ -s $OSG_TUNE_CACHE ] || cp onnxstream_amd/tune/mi355x.txt $OSG_TUNE_CACHE
  timeout 600 python bench.py --config SDXL --steps-per-image 10 --steps 20 --warmup 2 --cpu-passes 0 --windows 2 > ${T}_bench_sdxl_tune.json 2> ${T}_bench_sdxl_tune.err; line ${T}_bench_sdxl_tune.json "SDXL (tuning run)"
  rm -rf /tmp/prof_sdxl
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sdxl -o sdxl -- python bench.py --config SDXL --steps-per-image 10 --steps 20 --warmup 2 --cpu-passes 0 --profile-reps 1 --windows 0 > ${T}_rocprof_sdxl.log 2>&1
  echo "rocprofv3 (SDXL) exit $?"
  for f in $(find /tmp/prof_sdxl -name "*kernel_stats.csv"); do cp $f ${T}_sdxl_rocprofv3_kernel_stats.csv; done
  python tools/graph_trace.py $(find /tmp/prof_sdxl -name "*kernel_trace.csv" | head -1) > ${T}_sdxl_graph_timeline.txt 2>&1; head -30 ${T}_sdxl_graph_timeline.txt; unset OSG_TUNE_CACHE ;;
*) echo "unknown recipe $R" ;;
esac; done
