#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc run (counter_collection.csv [+ kernel_trace.csv]) into one row per kernel:
dispatch count, average duration and the average of every counter per dispatch.

    python tools/summarize_pmc.py gpurun_out/pmc1/<host>/<pid> > profiles/r01/pmc_sq_wait.csv
"""
import collections
import csv
import glob
import os
import sys


def short(name):
    return name.split("(")[0].replace("void ", "")


def main(run_dir):
    cc = glob.glob(os.path.join(run_dir, "*counter_collection.csv"))
    if not cc:
        raise SystemExit("no *counter_collection.csv under %s" % run_dir)
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(cc[0])):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    dur = collections.defaultdict(list)
    for path in glob.glob(os.path.join(run_dir, "*kernel_trace.csv")):
        for r in csv.DictReader(open(path)):
            dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    counters = sorted({c for v in agg.values() for c in v})
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "dispatches", "avg_duration_ns"] + ["%s_per_dispatch" % c for c in counters])
    for k in sorted(agg, key=lambda k: -sum(dur.get(k, [0]))):
        n = len(disp[k])
        avg = sum(dur[k]) / len(dur[k]) if dur.get(k) else 0
        w.writerow([k, n, int(avg)] + [int(agg[k].get(c, 0.0) / n) for c in counters])


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else ".")
