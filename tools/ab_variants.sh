#!/bin/bash
# A/B of compile-time variants on the GPU box (hipcc is in the image): for every argument -- a string of -D flags, "" for the
# default build -- rebuild the kernel files with it, run a quick parity subset, then print bench figures for the given configurations.
#   usage: CFGS="M C3" tools/ab_variants.sh "" "-DSRLA_FFT_SWZ12" "-DSRLA_TW_DERIVE"
CFGS="${CFGS:-M C3}"
REPS="${REPS:-2}"
for v in "$@"; do
  touch srla_amd/csrc/kernels_common.h
  make -s -C srla_amd/csrc EXTRA="$v" -j8 2>&1 | grep -E "error" | head -5
  if [ "${PARITY:-1}" = "1" ]; then
    python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or item_records or full_scale" 2>&1 | tail -1
  fi
  for c in $CFGS; do
    for rep in $(seq $REPS); do
      line=$(python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1)
      python - "$c" "[$v]" "$line" <<'P'
import json, sys
c, env, line = sys.argv[1:4]
try:
    d = json.loads(line)
    st = d["roofline"]["stages"]
    print("%-3s %-34s value %8.1f resident %8s  stages(ms/job): %s  60s %s 10s %s" % (c, env, d["value"], (d.get("device_resident") or {}).get("value"),
          " ".join("%s=%.3f" % (k.replace("srla_", ""), v["ms_per_job"]) for k, v in st.items()),
          (d.get("stream_60s") or {}).get("value"), (d.get("stream_10s") or {}).get("value")), flush=True)
except Exception as e:
    print(c, env, "FAILED", e, line[:300], flush=True)
P
    done
  done
done
# leave the default build behind
touch srla_amd/csrc/kernels_common.h; make -s -C srla_amd/csrc -j8 2>&1 | grep -E "error" | head -5
