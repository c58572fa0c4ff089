#!/usr/bin/env python3
"""Kernel timeline of a few steps from a rocprofv3 --kernel-trace CSV: per step (delimited by the masks_kernel launch that opens it) the span,
the time with NO kernel running (dependency / launch gaps), with exactly one, with two or more; the largest idle gaps and what ran on either
side of them.   python tools/timeline_gaps.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
starts = [i for i, e in enumerate(ev) if "masks_kernel" in e[2]]
if len(starts) < 3:
    raise SystemExit("fewer than three steps in the trace")
short = lambda n: n.replace("void ", "").replace("wsl::", "").split("(")[0][:44]   # noqa: E731
tot = {"span": 0.0, "idle": 0.0, "one": 0.0, "two": 0.0}
gaps = []
steps = list(zip(starts[1:-1], starts[2:]))            # whole steps only, the first one dropped
for a, b in steps:
    seg = ev[a:b]
    t0, t1 = seg[0][0], ev[b][0]
    pts = []
    for s, e, _ in seg:
        pts.append((s, 1))
        pts.append((min(e, t1), -1))
    pts.sort()
    depth, last = 0, t0
    for t, d in pts:
        dt = t - last
        if dt > 0:
            tot["idle" if depth == 0 else "one" if depth == 1 else "two"] += dt
        depth += d
        last = t
    tot["span"] += t1 - t0
    # idle gaps: between the running maximum of the end times and the next start
    end_max, prev = seg[0][1], seg[0][2]
    for s, e, n in seg[1:]:
        if s > end_max:
            gaps.append((s - end_max, short(prev), short(n)))
        if e > end_max:
            end_max, prev = e, n
n = len(steps)
print(f"{n} whole steps: span {tot['span'] / n / 1e6:.3f} ms per step; no kernel running {tot['idle'] / n / 1e6:.3f} ms, exactly one "
      f"{tot['one'] / n / 1e6:.3f} ms, two or more {tot['two'] / n / 1e6:.3f} ms; {len(gaps) / n:.0f} idle gaps per step, mean "
      f"{sum(g[0] for g in gaps) / max(1, len(gaps)) / 1e3:.2f} us")
agg = {}
for g, a, b in gaps:
    k = (a, b)
    agg.setdefault(k, [0, 0.0])
    agg[k][0] += 1
    agg[k][1] += g
print("| idle between (kernel that ended last -> next kernel) | gaps per step | us per step |")
print("|---|---|---|")
for (a, b), (c, g) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"| {a} -> {b} | {c / n:.1f} | {g / n / 1e3:.1f} |")
