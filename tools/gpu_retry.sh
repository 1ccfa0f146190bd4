#!/bin/bash
# tools/gpu_retry.sh <log> <timeout> <command...>: one gpurun call; when no GPU slot is free (exit 3: nothing ran, nothing
# charged) wait and ask again, at most 8 times.  Any other outcome is final: a command that ran is never run twice.
log=$1; shift; to=$1; shift
for attempt in 1 2 3 4 5 6 7 8; do
  /usr/local/graft/bin/gpurun --timeout "$to" -- "$@" > "$log" 2>&1
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 120
done
exit 3
