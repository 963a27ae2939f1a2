#!/bin/bash
# round 6, GPU call 13: more buffer sets = more host run-ahead (SRLA_MI355X_SLOTS), host -> host at M and C2
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_exp13; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
 for c in M C2; do
  for slots in 5 6 7 9; do
   line=$(SRLA_MI355X_SLOTS=$slots timeout 300 python bench.py --config $c --steps 6 --warmup 2 --no-cpu-baseline --no-config-legs --no-extras 2>/dev/null | grep '^{' | tail -1)
   python - "$c" "$slots" "$line" >> $O/summary.txt <<'P'
import json, sys
c, v, line = sys.argv[1:4]
try:
    d = json.loads(line); ph = d["phase_ms_per_step"]
    print("%s SLOTS=%s value %8.1f  ms/step %.1f  wide busy %.1f  enqueue_host %.1f total_host %.1f" % (c, v, d["value"], d["ms_per_step"], ph["autocorr"] + ph["residual_cost"], ph["enqueue_host"], ph["total_host"]))
except Exception as e:
    print(c, v, "FAILED", e, line[:200])
P
  done
 done
done
cat $O/summary.txt
