#!/usr/bin/env python3
"""ordered kernel list (start offset, duration, gap before) of the last kernels in a rocprofv3 kernel trace CSV"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  dur %8.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r["Kernel_Name"][:90]))
    prev_end = max(prev_end, e)
