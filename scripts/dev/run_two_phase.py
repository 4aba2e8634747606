#!/usr/bin/env python3
"""development aid: the two-phase contract (symbolic + numeric) a few times on one workload, no checks (for --pmc passes)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spada_sim_amd as S
W = {"webbase": (S.GEN_WEBBASE_LIKE, 0, 0, 12347), "cop20k": (S.GEN_COP20K_LIKE, 0, 0, 12346), "rmat16": (S.GEN_RMAT, 16, 16, 22)}
m = S.generate(*W[sys.argv[1] if len(sys.argv) > 1 else "webbase"])
eng = S.Engine()
d = eng.upload(m)
for _ in range(3):
    eng.symbolic(d, d, 0, m.shape[0])
    eng.numeric_owned()
    print("numeric ms", eng.stats()["ms_numeric_call"])
eng.free(d)
eng.close()
