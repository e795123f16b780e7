#!/bin/bash
# Profiles the bench command on the GPU box: kernel-trace stats + PMC passes (each in a run of its own).
# Usage (through gpurun, from the repo root): bash tools/profile_round.sh r01b
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
timeout -s KILL 150 rocprofv3 --kernel-trace --stats -d $OUT/kt -o p --output-format csv -- $CMD > $OUT/kt.log 2>&1 || { echo "kernel trace failed"; tail -5 $OUT/kt.log; exit 1; }
pass() {
    name=$1; shift
    timeout -s KILL 150 rocprofv3 --pmc "$@" -d $OUT/pmc_$name -o p --output-format csv -- $CMD > $OUT/pmc_$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/pmc_$name.log; exit 1; }
}
pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY
pass sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS
pass occ SQ_LEVEL_WAVES SQ_ACCUM_PREV_HIRES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
pass write WRITE_SIZE
pass fetch FETCH_SIZE
grep -h '"metric"' $OUT/kt.log | tail -1 > $OUT/bench_line.json
echo "profile $TAG done"
