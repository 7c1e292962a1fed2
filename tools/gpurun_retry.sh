#!/bin/bash
# gpurun with a wait-and-retry on "no box / slot free" (exit 3: nothing ran, nothing charged) — never on a failed run
#   tools/gpurun_retry.sh LOGFILE [--timeout S] -- '<command>'
log=$1; shift
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  /usr/local/graft/bin/gpurun "$@" > "$log" 2>&1
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
