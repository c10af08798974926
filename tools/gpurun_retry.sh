#!/bin/bash
# gpurun with a host-side wait for a free slot: retries ONLY when the client reports that no box / slot was free (exit
# code 3: nothing ran, nothing was charged).  Usage: tools/gpurun_retry.sh <timeout> '<command>'
t=$1; shift
for attempt in 1 2 3 4 5 6 7 8; do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  echo "[gpurun_retry] no slot (attempt $attempt), waiting 150 s"
  sleep 150
done
exit 3
