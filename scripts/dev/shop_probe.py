"""development aid (GPU box, library built with -DSPADA_SHOP_PROBE): does the scatter's regime follow the memory of one workspace?  R-MAT 18, the stage behind the
plan (ms_cut) per one-pass call, with the workspaces named by SPADA_SHOP (bit mask: 1 scr_col, 2 scr_val, 4 part histograms, 8 parts, 16 range descriptors,
32 cut table, 64 tasks, 128 status words, 256 / 512 entry descriptors, 1024 row records, 2048 task counts, 4096 row accumulators) moved to new memory before
every call.  Without SPADA_SHOP (any build): the placement the engine chooses itself (SPADA_TRACE=1 prints it)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spada_sim_amd as S
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 18
m = S.generate(S.GEN_RMAT, scale, 16, 22)
eng = S.Engine(); d = eng.upload(m)
cap = S.count_products(m, m, 0, m.shape[0])
out = []
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    eng.fused_owned(d, d, 0, m.shape[0], cap)
    st = eng.stats()
    out.append(round(st["ms_cut"], 2))
print("SPADA_SHOP", os.environ.get("SPADA_SHOP"), "SPADA_PLACE", os.environ.get("SPADA_PLACE"), "ms behind the plan per call:", out, flush=True)
eng.free(d); eng.close()
