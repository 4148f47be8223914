#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes (one counter per run) over `bench.py --from-waveform` for the two log-mel kernels
# -> gpurun_out/<tag>_pmc_logmel.csv  (means per dispatch; FETCH_SIZE in KiB counts 64 B per 128-B request: double it, MI355X_MICROARCH.md)
tag=${1:-pmc}
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  out=gpurun_out/_pmclm_${tag}_$c
  rm -rf $out; mkdir -p $out
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out -- python3 bench.py --from-waveform --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_pmclm_$c.log 2>&1 || exit 1
done
python3 - $(find gpurun_out/_pmclm_${tag}_FETCH_SIZE gpurun_out/_pmclm_${tag}_WRITE_SIZE -name "*counter_collection.csv") > gpurun_out/${tag}_pmc_logmel.csv <<'PY'
import collections, csv, sys
acc = collections.defaultdict(list)
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        if "logmel" in r["Kernel_Name"]:
            import re
            acc[(re.search(r"logmel_\w+", r["Kernel_Name"]).group(0), r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
print("kernel,grid_size,counter,mean_per_dispatch_KiB,dispatches")
for (k, g, c), v in sorted(acc.items()):
    print(f"{k},{g},{c},{sum(v) / len(v):g},{len(v)}")
PY
rm -rf gpurun_out/_pmclm_${tag}_FETCH_SIZE gpurun_out/_pmclm_${tag}_WRITE_SIZE
cat gpurun_out/${tag}_pmc_logmel.csv
