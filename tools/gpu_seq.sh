#!/bin/bash
# Run GPU steps one after the other on the box; a step that TIMES OUT or dies by a signal ends the sequence (no further GPU step
# after a hang), an ordinary failure (exit 1: a failing test) does not.  usage: tools/gpu_seq.sh 'cmd 1' 'cmd 2' ...
# Every step runs under `timeout -k 10 ${STEP_TIMEOUT:-900}`.
mkdir -p gpurun_out
rc_all=0
for cmd in "$@"; do
    echo "=== $cmd"
    timeout -k 10 "${STEP_TIMEOUT:-900}" bash -c "$cmd"
    rc=$?
    echo "=== rc $rc"
    if [ $rc -ne 0 ]; then rc_all=$rc; fi
    if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then echo "=== stopping: timed out / killed"; exit $rc; fi
done
exit $rc_all
