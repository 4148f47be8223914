# per-kernel times of one tools/kbench.py mode under rocprofv3: bash tools/prof_kbench.sh <mode> [iters]  -> gpurun_out/prof_<mode>.txt
mode=$1; iters=${2:-20}
out=gpurun_out/prof_kb_$mode
rm -rf $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/kbench.py $mode --iters $iters > gpurun_out/prof_kb_$mode.log 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" > gpurun_out/prof_$mode.txt <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"]) / 1e3:10.1f} pct {r["Percentage"]}')
PY
cat gpurun_out/prof_$mode.txt
