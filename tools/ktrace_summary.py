#!/usr/bin/env python3
"""Kernel trace of the LAST encode call in a rocprofv3 --kernel-trace CSV: every dispatch with queue, start (ms from the call's
first dispatch) and duration, then per kernel the in-flight durations.   usage: ktrace_summary.py <kernel_trace.csv> [ndispatch]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last call starts at the last srla_or_reduce (device-resident probe) or after the largest idle gap
last = max((i for i, r in enumerate(rows) if "srla_or_reduce" in r["Kernel_Name"]), default=0)
sel = rows[last:]
t0 = int(sel[0]["Start_Timestamp"])
per = collections.defaultdict(list)
for r in sel:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    per[name].append((b - a) / 1e3)
    if "-q" not in sys.argv:
        print("%-36s q%-3s %8.3f %8.3f  grid %s" % (name[:36], r.get("Queue_Id"), (a - t0) / 1e6, (b - a) / 1e6, r["Grid_Size_X"]))
print("call: %.3f ms from first dispatch to last end" % ((max(int(r["End_Timestamp"]) for r in sel) - t0) / 1e6))
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    print("%-40s n=%-3d sum %8.1f us  each: %s" % (k[:40], len(v), sum(v), " ".join("%.0f" % x for x in v)))
