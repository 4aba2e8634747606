#!/bin/bash
# GPU box: task-kernel time of every libspada_abl<mask>.so next to the release library (scripts/probe_tasks.py <workloads>)
cd "$(dirname "$0")/../.."
echo "== release"; python scripts/probe_tasks.py "$@" 2>/dev/null | grep "one pass\|two phase"
for l in spada_sim_amd/lib/libspada_abl*.so; do
  echo "== $l"; SPADA_LIB_PATH=$PWD/$l python scripts/probe_tasks.py "$@" 2>/dev/null | grep "one pass\|two phase"
done
