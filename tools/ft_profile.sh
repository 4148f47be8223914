#!/bin/bash
# kernel trace of one fused fine-tune optimizer step -> gpurun_out/<tag>_finetune_by_shape.csv (top shapes by total time)
tag=${1:-ft}
export TMPDIR=/tmp
out=gpurun_out/_ftprof_$tag
rm -rf $out
rocprofv3 --kernel-trace --output-format csv -d $out -- python3 bench.py --mode finetune --steps 1 --warmup 1 --accum 8 > gpurun_out/${tag}_ftprof.log 2>&1
python3 tools/trace_by_shape.py $(find $out -name "*kernel_trace.csv" | head -1) > gpurun_out/${tag}_finetune_by_shape.csv
rm -rf $out
cat gpurun_out/${tag}_finetune_by_shape.csv
