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
# gaps of several milliseconds are host phases between the bench's timed region and its separately timed legs, not part of a step
big = [g for g in gaps if g[0] > 3e6]
gaps = [g for g in gaps if g[0] <= 3e6]
idle_small = sum(g[0] for g in gaps)
span = (t1 - t0) - sum(g[0] for g in big)
print("window %.3f ms (+ %d host phase(s) of %.1f ms between bench sections, excluded), device idle %.3f ms (%.1f %%)"
      % (span / 1e6, len(big), sum(g[0] for g in big) / 1e6, idle_small / 1e6, 100.0 * idle_small / span))
agg = {}
for g, a, b in gaps:
    k = (a, b)
    c = agg.setdefault(k, [0, 0])
    c[0] += g; c[1] += 1
for (a, b), (g, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]:
    print("%8.1f us total in %3d gaps (%.1f us each): after %-60s before %s" % (g / 1e3, c, g / c / 1e3, a, b))
