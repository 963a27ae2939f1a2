#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy (exit code 3): tools/gp.sh TIMEOUT 'command'
T=$1; shift
for try in 1 2 3 4 5 6 7 8 9 10; do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
