#!/bin/bash
# round 4, GPU call 9: the whole GPU suite on the current tree + rocprofv3 --kernel-trace --stats of the bench command (the command that faulted once at the end of round 3)
mkdir -p gpurun_out; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4c9_pytest_gpu.log 2>&1; tail -4 gpurun_out/r4c9_pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4c9_smoke.log 2>&1; tail -2 gpurun_out/r4c9_smoke.log
for rep in 1 2 3; do
  rm -rf /tmp/prof_f
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f -o fin -- python bench.py --steps 20 --warmup 2 --cpu-passes 0 --profile-reps 1 --windows 0 > gpurun_out/r4c9_rocprof_$rep.log 2>&1
  echo "rocprofv3 run $rep: exit $?"; grep -c "Memory access fault" gpurun_out/r4c9_rocprof_$rep.log; grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4c9_rocprof_$rep.log | head -1
done
for f in $(find /tmp/prof_f -name "*kernel_stats.csv"); do cp $f gpurun_out/r4c9_rocprof_kernel_stats.csv; done
T=$(find /tmp/prof_f -name "*kernel_trace.csv" | head -1)
python tools/graph_trace.py $T > gpurun_out/r4c9_graph_trace.txt 2>&1; head -30 gpurun_out/r4c9_graph_trace.txt
head -12 gpurun_out/r4c9_rocprof_kernel_stats.csv | cut -c1-200
