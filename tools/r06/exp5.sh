#!/bin/bash
# round 6, GPU call 5: is the wide stream busy?  kernel traces of one long call per configuration (device-resident input) -> tools/r06/wgaps.py
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_exp5; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for cfg in "1 0 4096 600" "2 0 4096 300" "2 3 4096 300" "2 3 8192 300"; do
  set -- $cfg
  tag=V$1_P$2_B$3
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/$tag -o run -- python $GRAFT_REPO_ROOT/tools/perf_probe.py $4 device 3 $1 $2 $3 > $GRAFT_REPO_ROOT/$O/$tag.log 2>&1
  echo "== $tag" >> $GRAFT_REPO_ROOT/$O/summary.txt
  python $GRAFT_REPO_ROOT/tools/r06/wgaps.py $GRAFT_REPO_ROOT/$O/$tag >> $GRAFT_REPO_ROOT/$O/summary.txt 2>&1
  tail -4 $GRAFT_REPO_ROOT/$O/$tag.log >> $GRAFT_REPO_ROOT/$O/summary.txt
  find $GRAFT_REPO_ROOT/$O/$tag -name '*.db' -delete; find $GRAFT_REPO_ROOT/$O/$tag -name '*agent_info*' -delete
  # keep the trace of the metric configuration only (size)
  [ "$tag" != "V1_P0_B4096" ] && rm -rf $GRAFT_REPO_ROOT/$O/$tag
done
cd $GRAFT_REPO_ROOT
SRLA_MI355X_TIMELINE=1 SRLA_MI355X_TIMING_STRIDE=1 timeout 120 python tools/perf_probe.py 600 device 2 1 0 4096 > $O/timeline_M.txt 2>&1
cat $O/summary.txt
