#!/bin/bash
# Dev tool (GPU box), round 6: parameterised runner for this round's gpurun calls (rounds 4-5: gpu_round.sh, whose recipes it reuses through `old <recipes> [args]`).
# usage: gpu_round6.sh <recipe>[,<recipe>...] [args]     output prefix gpurun_out/r06
#   floor4      tools/floor_probe4 (built here, travels): what a trivial node costs cold / hot / with its code prefetched
#   ref16       the reference's fp16 / fp32 outputs of the full-size nets on THIS host's CPU -> gpurun_out/ref16_fullsize_epyc.npz
#   old ...     tools/gpu_round.sh with ROUND=r06
mkdir -p gpurun_out; export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-.}
export ROUND=r06; T=gpurun_out/r06; export OSA_REQUIRE_ORACLE=1
IFS=',' read -ra RECIPES <<< "$1"; shift
for R in "${RECIPES[@]}"; do case $R in
floor4)
  timeout 300 tools/_build/floor_probe4 9 > ${T}_floor_probe4.txt 2>&1; echo "floor_probe4 exit $?"; cat ${T}_floor_probe4.txt ;;
ref16)
  timeout 1500 python tools/ref16_fullsize.py epyc gpurun_out ${1:-sd15,sd15_w8,sdxl,vae} > ${T}_ref16_epyc.log 2>&1; echo "ref16 exit $?"; tail -8 ${T}_ref16_epyc.log ;;
old)
  bash tools/gpu_round.sh "$@"; break ;;
*) echo "unknown recipe $R" ;;
esac; done
