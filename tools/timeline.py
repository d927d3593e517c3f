#!/usr/bin/env python3
"""Timeline of a pipelined run from a rocprofv3 kernel trace: per hardware queue the share of time spent inside
kernels and in the gaps between them, and how many of the frame's kernels run at once.
usage: timeline.py <kernel_trace.csv> [first_fraction last_fraction]"""
import csv, sys, collections
path = sys.argv[1]
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 0.6
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"].replace("void ","").replace("(anonymous namespace)::","").split("(")[0][:28]))
rows.sort()
ours = [r for r in rows if any(k in r[3] for k in ("insert", "resolve", "columns", "render", "fill_kernel", "frame"))]
n = len(ours)
sel = ours[int(n * lo):int(n * hi)]
t0, t1 = sel[0][0], sel[-1][1]
print(f"{len(sel)} dispatches over {(t1 - t0) / 1e3:.1f} us")
byq = collections.defaultdict(list)
for s, e, q, k in sel:
    byq[q].append((s, e, k))
for q, L in sorted(byq.items()):
    busy = sum(e - s for s, e, _ in L)
    gaps = collections.defaultdict(list)
    for (s0, e0, k0), (s1, e1, k1) in zip(L, L[1:]):
        gaps[f"{k0[:14]}->{k1[:14]}"].append(s1 - e0)
    span = L[-1][1] - L[0][0]
    print(f"queue {q}: {len(L)} kernels, busy {100 * busy / span:.0f} % of {span / 1e3:.0f} us")
    for g, v in sorted(gaps.items()):
        v.sort()
        print(f"    gap {g:32s} n={len(v):4d} median {v[len(v) // 2] / 1e3:6.2f} us  mean {sum(v) / len(v) / 1e3:6.2f}")
# concurrency: time-weighted histogram of the number of kernels running, and per kernel name
ev = []
for s, e, q, k in sel:
    ev.append((s, 1, k)); ev.append((e, -1, k))
ev.sort()
hist = collections.Counter(); cur = 0; last = ev[0][0]
running = collections.Counter(); pername = collections.Counter()
for t, d, k in ev:
    hist[cur] += t - last
    for kk, c in running.items():
        if c: pername[kk] += (t - last)
    last = t; cur += d; running[k] += d
tot = sum(hist.values())
print("kernels running at once:", {c: f"{100 * v / tot:.0f}%" for c, v in sorted(hist.items())})
print("share of wall time during which at least one instance runs:")
for k, v in pername.most_common():
    print(f"    {k:30s} {100 * v / tot:5.1f} %")
