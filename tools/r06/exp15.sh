#!/bin/bash
# round 6, GPU call 15: the library's own timeline of a host -> host call (where does the wide stream wait?)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_exp15; mkdir -p $O
export TMPDIR=/tmp
SRLA_MI355X_TIMELINE=1 SRLA_MI355X_TIMING_STRIDE=1 timeout 200 python tools/perf_probe.py 600 host 3 1 0 4096 > $O/timeline_M_host.txt 2>&1
SRLA_MI355X_TIMELINE=1 SRLA_MI355X_TIMING_STRIDE=1 SRLA_MI355X_SLOTS=7 timeout 200 python tools/perf_probe.py 600 host 3 1 0 4096 > $O/timeline_M_host_7sets.txt 2>&1
grep -c "" $O/*.txt
