#!/bin/bash
# Kernel trace of a 6-step bench.py run -> gpurun_out/<tag>_kernel_stats.csv, <tag>_kernel_by_shape.csv   (tools/profile_step.sh r3a)
tag=${1:-prof}; shift
out=gpurun_out/_prof_$tag
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/${tag}_bench_under_rocprof.log 2>&1
trace=$(find $out -name "*kernel_trace.csv" | head -1)
stats=$(find $out -name "*kernel_stats.csv" | head -1)
python3 tools/trace_by_shape.py "$trace" > gpurun_out/${tag}_kernel_by_shape.csv
head -25 "$stats" > gpurun_out/${tag}_kernel_stats.csv
python3 tools/trace_gaps.py "$trace" > gpurun_out/${tag}_trace_gaps.txt
rm -rf $out
tail -2 gpurun_out/${tag}_bench_under_rocprof.log | cut -c1-300
cat gpurun_out/${tag}_trace_gaps.txt
