#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes (one counter per run, as MI355X_MICROARCH.md prescribes) over bench.py -> gpurun_out/<tag>_pmc_gemm_pp.csv
# (extra environment, e.g. LA_LIB_PATH=... LA_GEMM_MBLOCK=0 for the experiment build's tile order A/B, is inherited by the profiled run)
tag=${1:-pmc}
counters=${2:-"FETCH_SIZE WRITE_SIZE"}
export TMPDIR=/tmp
files=""
for c in $counters; do
  out=gpurun_out/_pmc_${tag}_$c
  rm -rf $out; mkdir -p $out
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > gpurun_out/${tag}_pmc_$c.log 2>&1 || exit 1
  files="$files $(find $out -name '*counter_collection.csv')"
done
python3 tools/pmc_summary.py $files > gpurun_out/${tag}_pmc_gemm_pp.csv
for c in $counters; do rm -rf gpurun_out/_pmc_${tag}_$c; done
cat gpurun_out/${tag}_pmc_gemm_pp.csv
