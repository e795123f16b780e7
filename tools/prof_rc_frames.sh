#!/bin/bash
# kernel trace of the resident render-cache frame loop (1080p, one sample per 16 pixels, 60 frames): bash tools/prof_rc_frames.sh TAG
set -u
TAG=${1:-rcf}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
NODE=$(command -v node)
timeout -s KILL 150 rocprofv3 --kernel-trace --stats -d $OUT/kt -o p --output-format csv -- $NODE $R/path-tracer_amd/js/test/bench_rc_frames.mjs 1920 1080 16 60 1 > $OUT/kt.log 2>&1 || { echo "kernel trace failed"; tail -5 $OUT/kt.log; exit 1; }
echo "done"
