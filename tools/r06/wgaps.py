#!/usr/bin/env python3
"""How busy is the wide stream?  From a rocprofv3 --kernel-trace CSV: the union of the wide kernels' (srla_autocorr*, srla_residual_cost*)
execution intervals against the span from the first to the last of them, the gaps between consecutive wide kernels, and which kernels
ran during the gaps.      python tools/r06/wgaps.py <dir with *kernel_trace.csv> [skip_first_ms]"""
import collections
import csv
import glob
import sys

rows = []
for path in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Queue_Id", "?")))
rows.sort()
wide = [r for r in rows if r[2].startswith("srla_autocorr") or r[2].startswith("srla_residual_cost")]
if not wide:
    raise SystemExit("no wide kernels in the trace")
# the last call of the run (steady state): cut at the largest pause between wide kernels
cuts = [i for i in range(1, len(wide)) if wide[i][0] - wide[i - 1][1] > 3_000_000]
first = cuts[-1] if cuts else 0
wide = wide[first:]
t0, t1 = wide[0][0], max(w[1] for w in wide)
busy, cur_s, cur_e = 0, wide[0][0], wide[0][1]
gaps = []
for s, e, k, q in wide[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((cur_e, s))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = t1 - t0
print("wide kernels of the last call: %d launches, span %.3f ms, union busy %.3f ms = %.1f %%, %d gaps (%.3f ms)" % (
    len(wide), span / 1e6, busy / 1e6, 100.0 * busy / span, len(gaps), sum(b - a for a, b in gaps) / 1e6))
per = collections.defaultdict(lambda: [0, 0])
for s, e, k, q in wide:
    per[k][0] += 1
    per[k][1] += e - s
for k, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print("  %-48s n=%4d  sum %.3f ms  avg %.1f us" % (k[:48], n, t / 1e6, t / n / 1e3))
# what ran in the gaps
ing = collections.defaultdict(int)
for a, b in gaps:
    for s, e, k, q in rows:
        if e <= a or s >= b or k.startswith("srla_autocorr") or k.startswith("srla_residual_cost"):
            continue
        ing[k] += min(e, b) - max(s, a)
print("gap length histogram (us):", sorted(round((b - a) / 1e3, 1) for a, b in gaps)[-12:], "... largest twelve")
for k, t in sorted(ing.items(), key=lambda kv: -kv[1])[:8]:
    print("  in the gaps: %-44s %.3f ms" % (k[:44], t / 1e6))
# overlap among wide kernels (two of them executing at once)
ov = 0
for i in range(1, len(wide)):
    ov += max(0, min(wide[i - 1][1], wide[i][1]) - wide[i][0])
print("overlap between consecutive wide kernels: %.3f ms" % (ov / 1e6))
