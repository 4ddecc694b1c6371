#!/bin/bash
# Runs the given steps ("name|seconds|command") one after the other on the GPU box, each under its own `timeout -k 10`, output
# to gpurun_out/<name>.log.  An ordinary failure (a red test) does not stop the sequence; a step that was KILLED at its limit
# (exit 124 / 137) does -- no further GPU step is started behind a hung one.
mkdir -p gpurun_out
for step in "$@"; do
    name="${step%%|*}"; rest="${step#*|}"; secs="${rest%%|*}"; cmd="${rest#*|}"
    echo "== $name (limit ${secs}s): $cmd"
    timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
    rc=$?
    echo "== $name rc=$rc"; tail -n 6 "gpurun_out/$name.log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "== $name was killed at its limit: stopping here"; exit $rc; fi
done
exit 0
