#!/bin/bash
# Builds the library of the last commit into path-tracer_amd/lib/libptr_hip_prev.so (for tools/ab3.sh against the working tree).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
rm -rf /tmp/ptr_prev; git -C $R worktree prune
git -C $R worktree add -f /tmp/ptr_prev HEAD -q
mkdir -p /tmp/ptr_prev/path-tracer_amd/lib
make -C /tmp/ptr_prev/path-tracer_amd/csrc -j8 > /tmp/ptr_prev_build.log 2>&1 || { tail -5 /tmp/ptr_prev_build.log; exit 1; }
cp /tmp/ptr_prev/path-tracer_amd/lib/libptr_hip.so $R/path-tracer_amd/lib/libptr_hip_prev.so
git -C $R worktree remove --force /tmp/ptr_prev
echo "built libptr_hip_prev.so from $(git -C $R rev-parse --short HEAD)"
