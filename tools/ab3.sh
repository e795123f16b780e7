#!/bin/bash
# A/B/C... timing of several builds of libptr_hip.so on the SAME box: bash tools/ab3.sh libX.so libY.so ... -- [prof_run.py arguments]
# (the in-tree library is restored at the end); three rounds, alternating
set -u
LIBS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done
[ $# -gt 0 ] && shift
L=path-tracer_amd/lib/libptr_hip.so
cp $L /tmp/lib_intree.so
for round in 1 2 3; do
    for B in "${LIBS[@]}"; do
        cp $B $L; echo -n "$(basename $B): "; timeout -k 10 100 python tools/prof_run.py "$@" 2>/dev/null | sed 's/.*\(([0-9.]*, [0-9.]*)\)\].*wall ms/last kernel ms \1  wall ms/' || exit 1
    done
done
cp /tmp/lib_intree.so $L
