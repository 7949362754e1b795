#!/bin/bash
# usage: gpu_when_free.sh TIMEOUT_S 'command'  -- runs one gpurun call; if no slot is free (exit 3: nothing ran, nothing
# charged) waits two minutes and asks again, up to 20 times.  Any other exit status ends it.
t=$1; shift
for k in $(seq 1 20); do
  gpurun --timeout "$t" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 120
done
exit 3
