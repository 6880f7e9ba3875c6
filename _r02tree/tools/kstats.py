#!/usr/bin/env python3
"""print the top kernels of a rocprofv3 --kernel-trace --stats CSV: calls per update, average us"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
upd = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("sum of kernel time per update: %.3f ms" % (tot / upd / 1e6))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 16]:
    print("%-74s calls/upd %5.1f avg %8.1f us" % (r["Name"][:74], int(r["Calls"]) / upd, float(r["AverageNs"]) / 1e3))
