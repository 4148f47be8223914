"""rocprofv3 kernel trace CSV -> per (kernel, grid, workgroup) summary: calls, average / minimum us, total ms."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "")
    key = (name[:70], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"])
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg.setdefault(key, [0, 0, 1 << 62])
    a[0] += 1; a[1] += d; a[2] = min(a[2], d)
w = csv.writer(sys.stdout)
w.writerow(["kernel", "grid_x", "grid_y", "grid_z", "wg_x", "vgpr", "agpr", "lds", "calls", "avg_us", "min_us", "total_ms"])
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    w.writerow(list(k) + [a[0], round(a[1] / a[0] / 1e3, 1), round(a[2] / 1e3, 1), round(a[1] / 1e6, 2)])
