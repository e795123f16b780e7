#!/bin/bash
# Profiles the bench command on the GPU box: kernel-trace stats + PMC passes (each in a run of its own).
# Usage (through gpurun, from the repo root): bash tools/profile_round.sh r03
# The kernel trace runs the bench as the driver does (many steps: the average duration is the steady state's, the first cold
# launches weigh little); the counter passes run few steps (a PMC run serialises and slows every dispatch).
# Second target: the many-sphere kernel (BASELINE config 5, tools/prof_run.py ... many), instruction mix + occupancy passes.
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
KT="python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline"
CMD="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
MANY="python3 $R/tools/prof_run.py 3 4 hybrid many"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $OUT/kt -o p --output-format csv -- $KT > $OUT/kt.log 2>&1 || { echo "kernel trace failed"; tail -5 $OUT/kt.log; exit 1; }
echo "kernel trace done"
pass() {
    name=$1; shift
    timeout -s KILL 150 rocprofv3 --pmc "$@" -d $OUT/pmc_$name -o p --output-format csv -- $CMD > $OUT/pmc_$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/pmc_$name.log; exit 1; }
    echo "pass $name done"
}
pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY
pass sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS
# occupancy: SQ_LEVEL_WAVES / SQ_ACCUM_PREV_HIRES read 0 under rocprofv3 on gfx950 (and so do the derived occupancy metrics);
# waves per SIMD = SQ_WAVE_CYCLES / SQ_BUSY_CU_CYCLES, calibrated by tools/ubench/occupancy_cal.hip
pass occ SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE
pass write WRITE_SIZE
pass fetch FETCH_SIZE
for name in sq1 occ; do
    case $name in
        sq1) C="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY";;
        occ) C="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT";;
    esac
    timeout -s KILL 150 rocprofv3 --pmc $C -d $OUT/pmc_many_$name -o p --output-format csv -- $MANY > $OUT/pmc_many_$name.log 2>&1 || { echo "pass many_$name failed"; tail -5 $OUT/pmc_many_$name.log; exit 1; }
    echo "pass many_$name done"
done
grep -h '"metric"' $OUT/kt.log | tail -1 > $OUT/bench_line.json
echo "profile $TAG done"
