#!/bin/bash
# A/B of environment settings with extra bench arguments on one box, interleaved: tools/ab_args.sh "BENCH ARGS" ROUNDS "ENV1" "ENV2" ...
ARGS=$1; ROUNDS=$2; shift 2
for r in $(seq 1 $ROUNDS); do
  for env in "$@"; do
    line=$(env $env python bench.py $ARGS --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1)
    python - "$ARGS" "$env" "$line" <<'P'
import json, sys
a, env, line = sys.argv[1:4]
try:
    d = json.loads(line)
    print("%-24s %-28s value %8.1f  60s %s  10s %s | %s" % (a, env, d["value"], d.get("stream_60s", {}).get("value"),
          d.get("stream_10s", {}).get("value"), d.get("host_buffers")), flush=True)
except Exception as e:
    print(a, env, "FAILED", e, line[:300], flush=True)
P
  done
done
