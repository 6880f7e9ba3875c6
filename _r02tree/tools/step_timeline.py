#!/usr/bin/env python3
"""kernel timeline of ONE full bench step (rollout -> next rollout) from a rocprofv3 kernel trace CSV:
start offset, duration, stream, name.  usage: step_timeline.py trace.csv [which rollout (from the end), default 3]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ro = [i for i, r in enumerate(rows) if "rollout_kernel" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
a, b = ro[-k - 1], ro[-k]
t0 = int(rows[a]["Start_Timestamp"])
busy_end = t0
idle = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > busy_end:
        idle += s - busy_end
    busy_end = max(busy_end, e)
    n = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    print("+%8.1f us  %8.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?")[-3:], n[:70]))
print("step %.1f us, device idle inside it %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3, idle / 1e3))
