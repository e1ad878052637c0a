#!/bin/bash
# gpuq.sh TIMEOUT 'command' -- gpurun, waiting for a free GPU slot: retried ONLY on exit code 3 (no slot / box
# free right now, nothing ran, nothing charged).  Any other result is returned as it is.
T=$1; shift
for i in $(seq 1 30); do
    /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
    rc=$?
    if [ $rc -ne 3 ]; then exit $rc; fi
    sleep 45
done
exit 3
