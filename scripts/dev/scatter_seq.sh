#!/bin/bash
# development aid (GPU box): what k_big_scatter costs when its stores are coalesced (a measurement build, -DSPADA_SCATTER_SEQ: stores in
# walk order, results wrong, the run aborted behind the scatter) against the product's scatter, on chunks of R-MAT 22; durations by rocprofv3
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r05/scatter_seq
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
cat > /tmp/seq_probe.py <<PY
import os, sys
sys.path.insert(0, "$REPO")
import spada_sim_amd as S
m = S.generate(S.GEN_RMAT, 22, 16, 22)
bounds = S.partition_rows(m, m, 69)
eng = S.Engine(); d = eng.upload(m)
for c in (0, 8, 34, 68):
    for rep in range(2):
        try:
            eng.symbolic(d, d, int(bounds[c]), int(bounds[c + 1]))
        except Exception as e:
            print("chunk", c, "stopped:", str(e)[:80])
PY
for v in prod seq; do
  [ $v = seq ] && export SPADA_LIB_PATH=$REPO/spada_sim_amd/lib/dev_seq/libspada_spgemm.so
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/$v -o t -- python3 /tmp/seq_probe.py > $OUT/$v.log 2>&1
  python3 - <<PY
import csv
rows = [r for r in csv.DictReader(open("$OUT/$v/t_kernel_trace.csv")) if "k_big_scatter" in r["Kernel_Name"] or "k_big_hist" in r["Kernel_Name"]]
for k in ("k_big_hist", "k_big_scatter"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if k in r["Kernel_Name"]]
    print("$v", k, "ms per launch:", [round(x, 2) for x in d])
PY
done
