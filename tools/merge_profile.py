"""Attach the top kernels of a rocprofv3 --kernel-trace --stats run to a JSON record written by one of the bench tools.
usage: python tools/merge_profile.py <record.json> <dir with *_kernel_stats.csv> [note]"""
import csv, glob, json, os, sys

rec = json.load(open(sys.argv[1]))
paths = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_stats.csv"), recursive=True)
rows = []
for p in paths:
    for r in csv.DictReader(open(p)):
        rows.append({"kernel": r["Name"].replace("(anonymous namespace)::", "")[:90], "calls": int(r["Calls"]),
                     "total_ms": round(float(r["TotalDurationNs"]) / 1e6, 3), "avg_us": round(float(r["AverageNs"]) / 1e3, 1),
                     "percent": round(float(r["Percentage"]), 2)})
rows.sort(key=lambda r: -r["total_ms"])
rec["rocprof_top_kernels"] = rows[:10]
if len(sys.argv) > 3:
    rec["rocprof_run"] = sys.argv[3]
json.dump(rec, open(sys.argv[1], "w"), indent=1)
print(json.dumps(rec["rocprof_top_kernels"][:4]))
