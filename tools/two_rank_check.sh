#!/bin/bash
# Dev tool (1-GPU box): the N = 2 flow of bench.py end to end -- torchrun, prompt scatter, rank-0 tuning + table hand-over, per-rank device loop,
# max-over-ranks timing, result gather -- with both ranks on cuda:0 and gloo instead of RCCL (OSA_BENCH_ONE_GPU=1).  The number is meaningless
# (two ranks share one GPU); what is checked is that the path runs and both ranks agree on the plan.
mkdir -p gpurun_out
export OSA_BENCH_ONE_GPU=1
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 20 --warmup 2 --cpu-passes 0 --profile-reps 0 > gpurun_out/two_rank.json 2> gpurun_out/two_rank.err
echo "exit $?"; tail -5 gpurun_out/two_rank.err; cut -c1-400 gpurun_out/two_rank.json
