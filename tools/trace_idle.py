"""GPU busy / idle of a rocprofv3 kernel trace: python tools/trace_idle.py <kernel_trace.csv> [skip_fraction]
Sums kernel durations and the gaps between consecutive kernels (by start time, all streams merged) over the LAST optimizer step of the
trace (between the last two groups of adamw_kernel launches); lists the kernels that precede the largest gaps."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# one optimizer step = from the end of the second-to-last group of adamw_kernel launches to the end of the last one
ad = [e[1] for e in ev if "adamw" in e[2]]
groups = []
for t in ad:
    if not groups or t - groups[-1] > 50e6:
        groups.append(t)
    else:
        groups[-1] = t
if len(groups) < 2:
    sys.exit("fewer than two optimizer steps in the trace")
ev = [e for e in ev if groups[-2] < e[0] and e[1] <= groups[-1]]
busy, end, gaps = 0, ev[0][0], collections.Counter()
gap_total, n_small = 0, 0
for s, e, k in ev:
    if s > end:
        gap_total += s - end
        gaps[prev[:60]] += s - end
    busy += max(0, e - max(s, end))
    end = max(end, e)
    prev = k
    if e - s < 20000: n_small += 1
span = end - ev[0][0]
print(f"span {span / 1e6:.1f} ms, busy {busy / 1e6:.1f} ms ({100 * busy / span:.1f} %), gaps {gap_total / 1e6:.1f} ms, launches {len(ev)}, under 20 us: {n_small}")
for k, g in gaps.most_common(12):
    print(f"  idle after {k:60s} {g / 1e6:8.2f} ms")
by = collections.Counter(); cnt = collections.Counter()
for s, e, k in ev:
    by[k[:70]] += e - s; cnt[k[:70]] += 1
print("kernel time by name over the step:")
for k, t in by.most_common(45):
    print(f"  {k:70s} {t / 1e6:8.2f} ms {cnt[k]:6d} launches")
