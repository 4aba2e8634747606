#!/bin/bash
# development aid (GPU box): scripts/dev/ab.sh "<variants>" "<probe workloads>" [pytest args]  -- the parity tests once on the default
# library, then scripts/probe_tasks.py per variant library (spada_sim_amd/lib/libspada_<variant>.so, scripts/build_variant.sh)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
OUT=gpurun_out/ab/out.txt
: > $OUT
if [ -n "$3" ]; then timeout 1500 python -m pytest $3 -m gpu -x -q 2>&1 | tail -15 >> $OUT; fi
for v in $1; do
  export SPADA_LIB_PATH=$PWD/spada_sim_amd/lib/libspada_$v.so
  echo "#### $v" >> $OUT
  timeout 600 python scripts/probe_tasks.py $2 >> $OUT 2>&1
done
cat $OUT
