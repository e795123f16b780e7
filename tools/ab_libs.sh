#!/bin/bash
# A/B timing of two builds of libptr_hip.so on the SAME box (boxes differ by several per cent):
#   bash tools/ab_libs.sh path-tracer_amd/lib/libptr_hip_B.so [prof_run.py arguments]
# runs tools/prof_run.py alternately with the in-tree library (A) and the given one (B), three rounds.
set -u
B=$1; shift
L=path-tracer_amd/lib/libptr_hip.so
cp $L /tmp/libA.so
for round in 1 2 3; do
    cp /tmp/libA.so $L; echo -n "A: "; timeout -k 10 100 python tools/prof_run.py "$@" | sed 's/.*wall ms/wall ms/' || exit 1
    cp $B $L;           echo -n "B: "; timeout -k 10 100 python tools/prof_run.py "$@" | sed 's/.*wall ms/wall ms/' || exit 1
done
cp /tmp/libA.so $L
