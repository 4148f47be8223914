#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes (one counter per run, as MI355X_MICROARCH.md prescribes) over bench.py -> gpurun_out/<tag>_pmc_gemm_pp.csv
tag=${1:-pmc}
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  out=gpurun_out/_pmc_${tag}_$c
  rm -rf $out; mkdir -p $out
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_pmc_$c.log 2>&1 || exit 1
done
python3 tools/pmc_summary.py $(find gpurun_out/_pmc_${tag}_FETCH_SIZE gpurun_out/_pmc_${tag}_WRITE_SIZE -name "*counter_collection.csv") > gpurun_out/${tag}_pmc_gemm_pp.csv
rm -rf gpurun_out/_pmc_${tag}_FETCH_SIZE gpurun_out/_pmc_${tag}_WRITE_SIZE
cat gpurun_out/${tag}_pmc_gemm_pp.csv
