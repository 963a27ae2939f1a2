#!/bin/bash
# throughput of short streams through SRLAEncoder_EncodeWhole (pageable host memory -> pageable host memory): the settings are
# tried in turn, three rounds, so that clock ramps and host noise hit all of them alike; median of each round's 12 calls
# usage: tools/short_streams.sh "ENV=.." "ENV=.." ...
SECS="${SECS:-10 30 60}"
for round in 1 2 3; do
  for env in "$@"; do
    for secs in $SECS; do
      med=$(env $env python tools/perf_probe.py $secs host 12 2>/dev/null | awk '/rep/ {print $5}' | sort -n | sed -n 6p)
      echo "round $round  $env  ${secs}s  $med"
    done
  done
done
