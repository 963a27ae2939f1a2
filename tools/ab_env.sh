#!/bin/bash
# A/B of environment settings on one box, interleaved so that box drift cancels: tools/ab_env.sh CONFIG ROUNDS "ENV1" "ENV2" ...
CFG=$1; ROUNDS=$2; shift 2
for r in $(seq 1 $ROUNDS); do
  for env in "$@"; do
    line=$(env $env python bench.py --config $CFG --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1)
    python - "$CFG" "$env" "$line" <<'P'
import json, sys
c, env, line = sys.argv[1:4]
try:
    d = json.loads(line)
    print("%-3s %-40s value %8.1f  resident %8s  60s %s  10s %s" % (c, env, d["value"], d.get("device_resident", {}).get("value"),
          d.get("stream_60s", {}).get("value"), d.get("stream_10s", {}).get("value")), flush=True)
except Exception as e:
    print(c, env, "FAILED", e, line[:300], flush=True)
P
  done
done
