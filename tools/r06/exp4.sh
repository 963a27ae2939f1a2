#!/bin/bash
# round 6, GPU call 4: host / launch-shape experiments on ONE box, interleaved: the 4096-point class on 512 threads (AC_WIDE), srla_residual_cost on
# a stream of its own (RC_STREAM, with and without more hardware queues), srla_residual_cost<1,2> compiled for six wavefronts per SIMD
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_exp4; mkdir -p $O
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O2 -Wno-unused-value -o /tmp/valu_rates tools/probes/valu_rates.hip > $O/valu_build.log 2>&1 && /tmp/valu_rates > $O/valu_rates.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "options" > $O/parity.out 2>&1; tail -2 $O/parity.out > $O/summary.txt
run() {  # config, label, env...
  c=$1; label=$2; shift 2
  line=$(env "$@" timeout 300 python bench.py --config $c --steps 6 --warmup 2 --no-cpu-baseline --no-config-legs 2>/dev/null | grep '^{' | tail -1)
  python - "$c" "$label" "$line" >> $O/summary.txt <<'P'
import json, sys
c, label, line = sys.argv[1:4]
try:
    d = json.loads(line)
    st = d["roofline"]["stages"]
    print("%-3s %-26s value %8.1f resident %8s  stages(ms/job): %s  60s %s 10s %s" % (c, label, d["value"], (d.get("device_resident") or {}).get("value"),
          " ".join("%s=%.3f" % (k.replace("srla_", ""), v["ms_per_job"]) for k, v in st.items() if v.get("ms_per_job")),
          (d.get("stream_60s") or {}).get("value"), (d.get("stream_10s") or {}).get("value")), flush=True)
except Exception as e:
    print(c, label, "FAILED", e, line[:300], flush=True)
P
}
for rep in 1 2; do
 for c in M C3 C5; do
  run $c default X=1
  run $c ac_wide SRLA_MI355X_AC_WIDE=1
  run $c rc_stream SRLA_MI355X_RC_STREAM=1
  run $c rc_stream_8q SRLA_MI355X_RC_STREAM=1 GPU_MAX_HW_QUEUES=8
  run $c rc6_waves SRLA_PRODUCT_SO=$PWD/srla_amd/libsrla_rc6.so
  run $c ac_wide+rc_stream SRLA_MI355X_AC_WIDE=1 SRLA_MI355X_RC_STREAM=1
 done
done
SRLA_MI355X_AC_WIDE=1 bash tools/r06/alone_cfg.sh r06_exp4 srla_amd/libsrla_mi355x.so 2 3 4096 >> $O/summary.txt 2>&1
cat $O/summary.txt
