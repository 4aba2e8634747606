#!/usr/bin/env python3
"""development aid (GPU box): does the scatter phase's regime (profiles/r05_rmat22_runs.txt) belong to the ALLOCATION or to the moment?  One process, several
engines one after the other (fresh workspaces each), a few symbolic calls on one chunk of R-MAT 22 per engine: ms from the plan's end to the task kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spada_sim_amd as S

chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 34
m = S.generate(S.GEN_RMAT, 22, 16, 22)
bounds = S.partition_rows(m, m, 69)
b0, b1 = int(bounds[chunk]), int(bounds[chunk + 1])
for e in range(6):
    eng = S.Engine()
    d = eng.upload(m)
    cuts = []
    for rep in range(5):
        eng.symbolic(d, d, b0, b1)
        cuts.append(eng.stats()["ms_cut"])
    print(f"engine {e}: scatter phase of chunk {chunk}, five calls: " + " ".join(f"{c:.2f}" for c in cuts), flush=True)
    eng.free(d)
    eng.close()
    if e == 2:
        time.sleep(5.0)   # (an idle gap)
