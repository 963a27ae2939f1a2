#!/bin/bash
# round 6, GPU call 11: a call's last job stores its blocks itself (SRLA_MI355X_DIRECT_TAIL): parity, then short streams A/B; the default line with the metric's handle released before the legs
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_exp11; mkdir -p $O
export TMPDIR=/tmp
timeout 1000 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/suite.out 2> $O/suite.err; echo "suite rc=$?" > $O/summary.txt; tail -1 $O/suite.out >> $O/summary.txt
timeout 400 python tools/gpu_sweep.py 150 611 --mutate --paths > $O/sweep_611.out 2>&1; tail -1 $O/sweep_611.out >> $O/summary.txt
for rep in 1 2 3; do
 for v in 1 0; do
  line=$(SRLA_MI355X_DIRECT_TAIL=$v timeout 300 python bench.py --config M --steps 4 --warmup 2 --no-cpu-baseline --no-config-legs 2>/dev/null | grep '^{' | tail -1)
  python - "$v" "$line" >> $O/summary.txt <<'P'
import json, sys
v, line = sys.argv[1:3]
try:
    d = json.loads(line)
    print("M DIRECT_TAIL=%s value %8.1f resident %8s 60s %s (%.3f ms) 10s %s (%.3f ms)" % (v, d["value"], d["device_resident"]["value"], d["stream_60s"]["value"], d["stream_60s"]["ms_per_call"], d["stream_10s"]["value"], d["stream_10s"]["ms_per_call"]))
except Exception as e:
    print("M", v, "FAILED", e, line[:200])
P
 done
done
( time timeout 600 python bench.py > $O/default_line.json 2> $O/default_line.err ) 2> $O/default_line.time
python - $O/default_line.json >> $O/summary.txt <<'P'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("default line: M", d["value"], "60s", d["stream_60s"]["value"], "10s", d["stream_10s"]["value"], "resident", d["device_resident"]["value"])
for k, v in d["configs"].items(): print(" ", k, v["value"], "cpu", v["cpu_baseline"]["value"], "x", v["speedup_vs_cpu_1core"])
P
cat $O/summary.txt $O/default_line.time
