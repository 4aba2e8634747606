#!/bin/bash
# development aid (GPU box): quick parity subset on the default library, then scripts/probe_tasks.py per library variant
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/first
OUT=gpurun_out/first/out.txt
: > $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tasks.py -m gpu -x -q --timeout 240 -k "$1" 2>&1 | tail -25 >> $OUT
for v in $2; do
  export SPADA_LIB_PATH=$PWD/spada_sim_amd/lib/libspada_$v.so
  echo "#### $v" >> $OUT
  timeout 300 python scripts/probe_tasks.py $3 >> $OUT 2>&1
done
cat $OUT
