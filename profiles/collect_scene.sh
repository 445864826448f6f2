#!/bin/bash
# Kernel trace and counter passes of one of the other configurations' frames (run from the repo root through gpurun):
#   collect_scene.sh TAG NAME COMMAND...      e.g.  collect_scene.sh r03 terrain python3 tools/gpu_terrain.py 1920 1080 4
# Raw output lands in gpurun_out/prof_<tag>_<name>/; profiles/summarize_scene.py condenses it into
# profiles/<tag>_<name>_counters.json.  Every pass is a run of its own with its own time limit; the chain stops at the
# first failure.
set -e -o pipefail
TAG=$1; NAME=$2; shift 2
OUT=gpurun_out/prof_${TAG}_${NAME}
REPO=$(pwd)
mkdir -p $OUT
export TMPDIR=/tmp
CMD="$*"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/kt -o kt -- $CMD > $REPO/$OUT/kt.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $REPO/$OUT/pmc_sq -o pmc -- $CMD > $REPO/$OUT/pmc_sq.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $REPO/$OUT/pmc_fetch -o pmc -- $CMD > $REPO/$OUT/pmc_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $REPO/$OUT/pmc_write -o pmc -- $CMD > $REPO/$OUT/pmc_write.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $REPO/$OUT/pmc_l2 -o pmc -- $CMD > $REPO/$OUT/pmc_l2.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum TCP_TAGRAM2_REQ_sum TCP_TAGRAM3_REQ_sum --output-format csv -d $REPO/$OUT/pmc_l1 -o pmc -- $CMD > $REPO/$OUT/pmc_l1.log 2>&1
cd $REPO
python3 profiles/summarize_scene.py $TAG $NAME "$CMD"
