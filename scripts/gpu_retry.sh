#!/bin/bash
# usage: scripts/gpu_retry.sh <log> <timeout s> <command...>   — gpurun with retries while no slot / box is free (exit code 3)
LOG=$1; TO=$2; shift; shift
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout $TO -- "$@" > $LOG 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 60
done
exit 3
