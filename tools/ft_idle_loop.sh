# GPU busy / idle of one fine-tune optimizer step: bash tools/ft_idle.sh <tag>  -> gpurun_out/<tag>_ft_idle.txt
tag=${1:-ft}
export TMPDIR=/tmp
out=gpurun_out/_ftidle_$tag
rm -rf $out
rocprofv3 --kernel-trace --output-format csv -d $out -- python3 bench.py --mode finetune --steps 2 --warmup 1 --accum 8 --accum-mode loop > gpurun_out/${tag}_ftidle.log 2>&1
python3 tools/trace_idle.py $(find $out -name "*kernel_trace.csv" | head -1) > gpurun_out/${tag}_ft_idle.txt
rm -rf $out
cat gpurun_out/${tag}_ft_idle.txt
