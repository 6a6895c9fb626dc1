#!/bin/bash
# gpurun, waiting for a free GPU slot: re-submits ONLY while the client answers "no box or slot free right now" (exit code 3:
# nothing ran, nothing was charged).  Any other outcome -- the command ran, passed, failed or was refused -- is returned as is.
# usage: tools/gpurun_wait.sh <timeout seconds> '<command>'
t=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 45
done
exit 3
