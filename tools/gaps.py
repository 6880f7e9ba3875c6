#!/usr/bin/env python3
"""Device idle time between kernels from a rocprofv3 --kernel-trace CSV: the largest gaps of the last steps and what
surrounds them.   python tools/gaps.py <kernel_trace.csv> [n_last_kernels]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
rows = rows[-n:]
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
busy_end = int(rows[0]["End_Timestamp"])
gaps = []
idle = 0
for a, b in zip(rows, rows[1:]):
    s = int(b["Start_Timestamp"])
    if s > busy_end:
        gaps.append((s - busy_end, a["Kernel_Name"][:60], b["Kernel_Name"][:60]))
        idle += s - busy_end
    busy_end = max(busy_end, int(b["End_Timestamp"]))
print("window %.3f ms, device idle %.3f ms (%.1f %%)" % ((t1 - t0) / 1e6, idle / 1e6, 100.0 * idle / (t1 - t0)))
agg = {}
for g, a, b in gaps:
    k = (a, b)
    c = agg.setdefault(k, [0, 0])
    c[0] += g; c[1] += 1
for (a, b), (g, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]:
    print("%8.1f us total in %3d gaps (%.1f us each): after %-60s before %s" % (g / 1e3, c, g / c / 1e3, a, b))
