#!/bin/bash
# usage: tools/gpu_retry.sh <timeout-seconds> '<command>' : runs gpurun, retrying only while no box is free (exit code 3)
T=$1; shift
for i in $(seq 1 12); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 75
done
exit 3
