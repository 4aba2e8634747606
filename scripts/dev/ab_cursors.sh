#!/bin/bash
# development aid (GPU box): per-range scatter cursors on / off -- chunks of R-MAT 22 and the task tests
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r05/ab_cursors
mkdir -p $OUT; cd $REPO
timeout 900 python -m pytest tests/test_gpu_tasks.py -x -q -m gpu > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
for v in 1 0; do
  SPADA_RANGE_CURSORS=$v timeout 600 python3 scripts/probe_chunks.py 22 69 0 8 34 68 > $OUT/chunks_$v.log 2>&1
  echo "== range cursors $v"; grep "== chunk\|symbolic" $OUT/chunks_$v.log | cut -c1-420
done
