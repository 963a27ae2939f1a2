#!/bin/bash
# A/B on the GPU box: the bench line's value / device_resident / per-stage times for a few configurations, with an environment
# variable toggled.   usage: tools/ab_bench.sh "VAR=0" "VAR=1" [configs...]
A="$1"; B="$2"; shift 2
CFGS="${@:-M C3 C5}"
for c in $CFGS; do
  for env in "$A" "$B"; do
    for rep in 1 2; do
      line=$(env $env python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1)
      python - "$c" "$env" "$line" <<'P'
import json, sys
c, env, line = sys.argv[1:4]
try:
    d = json.loads(line)
    st = d["roofline"]["stages"]
    print("%-3s %-28s value %8.1f resident %8s  stages(ms/job): %s  60s %s 10s %s" % (c, env, d["value"], d.get("device_resident", {}).get("value"),
          " ".join("%s=%.3f" % (k.replace("srla_", ""), v["ms_per_job"]) for k, v in st.items() if v.get("ms_per_job")),
          d.get("stream_60s", {}).get("value"), d.get("stream_10s", {}).get("value")), flush=True)
except Exception as e:
    print(c, env, "FAILED", e, line[:300], flush=True)
P
    done
  done
done
