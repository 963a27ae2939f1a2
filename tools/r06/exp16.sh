#!/bin/bash
# round 6, GPU call 16: smaller jobs for the -V 2 configurations (a 300 s call is 3.4 jobs of 4 Mi instants: mostly pipeline fill and drain)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_exp16; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
 for c in C3 C4 C5 M; do
  for js in 4194304 2097152 1048576; do
   line=$(SRLA_MI355X_JOB_SAMPLES=$js timeout 300 python bench.py --config $c --steps 6 --warmup 2 --no-cpu-baseline --no-config-legs 2>/dev/null | grep '^{' | tail -1)
   python - "$c" "$js" "$line" >> $O/summary.txt <<'P'
import json, sys
c, v, line = sys.argv[1:4]
try:
    d = json.loads(line)
    print("%s JOB_SAMPLES=%-8s value %8.1f resident %8s 60s %s 10s %s" % (c, v, d["value"], (d.get("device_resident") or {}).get("value"), (d.get("stream_60s") or {}).get("value"), (d.get("stream_10s") or {}).get("value")))
except Exception as e:
    print(c, v, "FAILED", e, line[:200])
P
  done
 done
done
cat $O/summary.txt
