#!/bin/bash
# Socket power and shader clock while bench.py runs (200 timed steps) -> gpurun_out/<tag>_power_samples.txt
# (rocm-smi reads; no settings are changed).  The first and last samples are taken with the GPU idle.
tag=${1:-pw}
out=gpurun_out/${tag}_power_samples.txt
sample() { rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Power|sclk|mclk|GPU use" | tr -s ' ' | tr '\n' ';'; echo; }
{
echo "# idle"; sample
python3 bench.py --steps 400 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_power_bench.json 2> gpurun_out/${tag}_power_bench.err &
pid=$!
sleep 14          # model build + head fit
for i in $(seq 1 24); do echo "# t=$i"; sample; sleep 0.5; done
wait $pid
echo "# idle again"; sleep 2; sample
} > $out 2>&1
cat $out | cut -c1-400
python3 -c "import json; d=json.load(open('gpurun_out/${tag}_power_bench.json')); print('ms_per_step', d['ms_per_step'])"
