#!/bin/bash
# round 6, GPU call 7: one pipeline step more between the solve chain and srla_residual_cost (SRLA_MI355X_C_SKEW), A/B on one box
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_exp7; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "options" > $O/parity.out 2>&1; tail -2 $O/parity.out > $O/summary.txt
run() {
  c=$1; label=$2; extra=$3; shift 3
  line=$(env "$@" timeout 300 python bench.py --config $c --steps 6 --warmup 2 --no-cpu-baseline --no-config-legs $extra 2>/dev/null | grep '^{' | tail -1)
  python - "$c" "$label" "$line" >> $O/summary.txt <<'P'
import json, sys
c, label, line = sys.argv[1:4]
try:
    d = json.loads(line)
    st = d["roofline"]["stages"]
    print("%-3s %-22s value %8.1f resident %8s  stages(ms/job): %s  60s %s 10s %s" % (c, label, d["value"], (d.get("device_resident") or {}).get("value"),
          " ".join("%s=%.3f" % (k.replace("srla_", ""), v["ms_per_job"]) for k, v in st.items() if v.get("ms_per_job")),
          (d.get("stream_60s") or {}).get("value"), (d.get("stream_10s") or {}).get("value")), flush=True)
except Exception as e:
    print(c, label, "FAILED", e, line[:300], flush=True)
P
}
for rep in 1 2 3; do
  run M default "" X=1
  run M c_skew "" SRLA_MI355X_C_SKEW=1
  run M c_skew_6sets "" SRLA_MI355X_C_SKEW=1 SRLA_MI355X_SLOTS=6
done
for rep in 1 2; do
 for c in C2 C3; do
  run $c default "" X=1
  run $c c_skew "" SRLA_MI355X_C_SKEW=1
 done
 run C5 default "" X=1
 run C5 c_skew_6sets "" SRLA_MI355X_C_SKEW=1 SRLA_MI355X_SLOTS=6
done
cat $O/summary.txt
