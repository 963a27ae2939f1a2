#!/bin/bash
# host-input 600 s encodes with the library's timeline: per setting the sorted call times and where the first job's stages sat
for env in "$@"; do
  env $env SRLA_MI355X_TIMELINE=1 SRLA_MI355X_TIMING_STRIDE=1 python tools/perf_probe.py 600 host 8 > /tmp/lp.txt 2>&1
  echo "$env: $(grep '^rep' /tmp/lp.txt | tail -7 | awk '{print $3}' | sort -n | tr '\n' ' ')"
  awk '/1 stream\(s\)/{f=0} /job of/ && !f {l=$0; f=1} END{print "   " l}' /tmp/lp.txt
done
