#!/bin/bash
# usage: scripts/gpu_retry.sh <log> <timeout-seconds> '<command>'   -- retries while no GPU slot is free (exit 3)
log=$1; to=$2; shift 2
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$to" -- "$@" > "$log" 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then echo "gpurun exit $rc" >> "$log"; exit $rc; fi
  sleep 45
done
echo "gave up" >> "$log"
