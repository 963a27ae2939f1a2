#!/usr/bin/env python3
"""fills DESIGN.md's @@X_VALUE@@ ... placeholders from profiles/r06/<config>/bench_line.json (run once, after tools/profiles.py summarize)"""
import json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = open(os.path.join(ROOT, "DESIGN.md")).read()
fmt = lambda v: "—" if v is None else ("{:,.0f}".format(v).replace(",", " "))
for c in ("M", "C2", "C3", "C4", "C5"):
    d = json.load(open(os.path.join(ROOT, "profiles", "r06", c, "bench_line.json")))
    vals = {"VALUE": d["value"], "RES": (d.get("device_resident") or {}).get("value"), "60": (d.get("stream_60s") or {}).get("value"), "10": (d.get("stream_10s") or {}).get("value")}
    for k, v in vals.items():
        s = s.replace("@@%s_%s@@" % (c, k), fmt(v))
open(os.path.join(ROOT, "DESIGN.md"), "w").write(s)
print(re.findall(r"@@\w+@@", s))
