#!/bin/bash
# development aid (GPU box): the task tests, then three alternating rounds of probe_tasks.py (R-MAT 16 / 18, web) and probe_chunks.py (R-MAT 22) on the product library and
# on spada_sim_amd/lib/dev_old/libspada_spgemm.so (the build of the commit before: git stash; make; cp)
mkdir -p gpurun_out/r05; timeout 600 python -m pytest tests/test_gpu_tasks.py -x -q -m gpu > gpurun_out/r05/t.txt 2>&1; grep -E "passed|failed|rror" gpurun_out/r05/t.txt | tail -3; cd $GRAFT_REPO_ROOT; for rep in 1 2 3; do for v in prod old; do if [ $v = old ]; then export SPADA_LIB_PATH=$GRAFT_REPO_ROOT/spada_sim_amd/lib/dev_old/libspada_spgemm.so; else unset SPADA_LIB_PATH; fi; echo "### $v"; timeout 300 python scripts/probe_tasks.py rmat16 rmat18 webbase 2>&1 | grep "one pass\|two phase"; timeout 200 python3 scripts/probe_chunks.py 22 69 0 8 34 68 2>&1 | python3 -c "
import re, sys
for l in sys.stdin:
    if l.startswith('   symbolic'):
        g = lambda k: float(re.search('%s.: ([0-9.]+)' % k, l).group(1)); print('   chunk: symbolic %.2f (big %.2f cut %.2f task %.2f)' % (g('ms_symbolic_call'), g('ms_big_expand'), g('ms_cut'), g('ms_task')), end='')
    if l.startswith('   numeric'):
        print(' numeric %s' % re.search('ms_numeric_call.: ([0-9.]+)', l).group(1)[:6])
"; done; done
