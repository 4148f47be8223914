"""rocprofv3 kernel trace CSV -> idle time between consecutive kernels of each queue (stream), and the chip's union-busy time.
usage: python tools/trace_gaps.py <kernel_trace.csv>"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
byq = collections.defaultdict(list)
for r in rows:
    nm = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
    nm = nm.split("(")[0][:34] + ":" + r["Grid_Size_X"] if "Grid_Size_X" in r else nm[:40]
    byq[r.get("Queue_Id", "0")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm))
allk = sorted((s, e) for q in byq.values() for s, e, _ in q)
t0, t1 = allk[0][0], max(e for _, e in allk)
busy, cur_s, cur_e = 0, allk[0][0], allk[0][1]
for s, e in allk[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"span {(t1 - t0) / 1e6:.2f} ms, some kernel running {busy / 1e6:.2f} ms ({100.0 * busy / (t1 - t0):.1f} %)")
for q, ks in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    ks.sort()
    gaps = [ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1)]
    small = [g for g in gaps if 0 <= g < 50000]
    hist = collections.Counter(min(g // 1000, 20) for g in small)
    print(f"queue {q}: {len(ks)} kernels, kernel time {sum(e - s for s, e, _ in ks) / 1e6:.2f} ms, gaps < 50 us: {len(small)} totalling {sum(small) / 1e6:.3f} ms "
          f"(median {sorted(small)[len(small) // 2] / 1e3 if small else 0:.1f} us), overlapping successors {sum(1 for g in gaps if g < 0)}")
    print("   gap histogram (us: count):", {k: hist[k] for k in sorted(hist)})

# gaps by (predecessor -> successor) kernel on the busiest queue
q, ks = max(byq.items(), key=lambda kv: len(kv[1]))
ks.sort()
pair = collections.defaultdict(list)
for i in range(len(ks) - 1):
    g = ks[i + 1][0] - ks[i][1]
    if 0 <= g < 50000:
        pair[(ks[i][2], ks[i + 1][2])].append(g / 1e3)
print(f"queue {q}: gap after -> before (us): count, median")
for k, v in sorted(pair.items(), key=lambda kv: -len(kv[1]))[:14]:
    v.sort()
    print(f"   {k[0]:44s} -> {k[1]:44s} {len(v):4d}  {v[len(v) // 2]:6.1f}")
