#!/bin/bash
# development aid (GPU box): scripts/dev/ab_env.sh "<VAR=value ...>" "<probe workloads>"  -- scripts/probe_tasks.py once per
# environment setting on the default library
cd $GRAFT_REPO_ROOT
for e in $1; do
  echo "#### $e"
  env $e timeout 600 python scripts/probe_tasks.py $2 2>&1 | grep "^==\|one pass\|two phase"
done
