"""rocprofv3 --pmc counter_collection CSVs -> the per-(grid, variant, counter) means bench.py reads for roofline.traffic.
usage: python tools/pmc_summary.py <counter_collection.csv> [more.csv ...] > profiles/<name>.csv"""
import collections
import csv
import sys

acc = collections.defaultdict(list)
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if "gemm_pp_kernel" not in name:
            continue
        variant = "f32out" if "gemm_pp_kernel<true" in name or "ILb1E" in name else "bf16out"
        acc[(r["Grid_Size"], variant, r["Counter_Name"])].append(float(r["Counter_Value"]))
print("grid_size,variant,counter,mean_per_dispatch")
for (g, v, c), vals in sorted(acc.items()):
    print(f"{g},{v},{c},{sum(vals) / len(vals):g}")
