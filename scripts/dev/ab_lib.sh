#!/bin/bash
# development aid (GPU box): the product library against a variant library (SPADA_LIB_PATH) on chunks of R-MAT 22, same box, alternating
# usage: scripts/dev/ab_lib.sh <variant .so relative to the repo> [chunks ...]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
VAR=$REPO/$1; shift
CH=${@:-0 8 34 68}
cd $REPO
for rep in 1 2; do for v in prod var; do
  if [ $v = var ]; then export SPADA_LIB_PATH=$VAR; else unset SPADA_LIB_PATH; fi
  timeout 600 python3 scripts/probe_chunks.py 22 69 $CH 2>&1 | python3 -c "
import re, sys
for l in sys.stdin:
    if l.startswith('   symbolic'):
        g = lambda k: float(re.search(\"'%s': ([0-9.]+)\" % k, l).group(1))
        print('$v', 'symbolic %.2f ms = stats %.2f + big %.2f + cut/scatter %.2f + task %.2f' % (g('ms_symbolic_call'), g('ms_row_stats'), g('ms_big_expand'), g('ms_cut'), g('ms_task')))
"
done; done
