#!/bin/bash
# usage (GPU box): scripts/collect_sq.sh <workload> <tag> [extra bench.py flags, e.g. --two-phase]
# Two separate --pmc passes of SQ counters (8 SQ slots per pass: MI355X_MICROARCH.md, rocprofv3 PMC slots; counters only, no trace
# domains) over `bench.py --steps 5 --warmup 2 --no-cpu-baseline`; per kernel and STEADY launch (scripts/pmc_per_launch.py: median, aborted launches dropped) ->
# gpurun_out/sq_<tag>/<tag>.json, to be copied into profiles/.
set -e
WL=${1:-webbase-1M}
TAG=${2:-r06_sq_$WL}
shift 2 || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --workload $WL $*"
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES \
    --output-format csv -d $OUT/p1 -o p -- $CMD > $OUT/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS \
    --output-format csv -d $OUT/p2 -o p -- $CMD > $OUT/p2.log 2>&1
python3 $REPO/scripts/pmc_per_launch.py $OUT/$TAG.json "rocprofv3 --pmc, two passes over: $CMD.  SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_INSTS_* count wave-instructions." $OUT/p1 $OUT/p2
