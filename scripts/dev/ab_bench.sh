#!/bin/bash
# development aid (GPU box): scripts/dev/ab_bench.sh "<variants>" <bench args ...>  -- one bench line per variant library
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/abb
V=$1; shift
for v in $V; do
  export SPADA_LIB_PATH=$PWD/spada_sim_amd/lib/libspada_$v.so
  python bench.py "$@" > gpurun_out/abb/$v.json 2> gpurun_out/abb/$v.err
  python - <<PY
import json
d = json.load(open("gpurun_out/abb/$v.json"))
k = d.get("roofline", {}).get("kernels") or []
print("$v", round(d["ms_per_step"], 2), [round(x["ms"], 1) for x in k])
PY
done
