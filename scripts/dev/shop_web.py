"""development aid (GPU box, library built with -DSPADA_SHOP_PROBE): does the one-pass task kernel's time follow the memory of one workspace?  k_task per call on the
web input with the workspaces named by SPADA_SHOP (bit mask as in shop_probe.py: 64 tasks, 128 status words, 256 / 512 entry descriptors, 1024 row records, ...) moved
before every call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spada_sim_amd as S
name = sys.argv[1] if len(sys.argv) > 1 else "webbase"
W = {"webbase": (S.GEN_WEBBASE_LIKE, 0, 0, 12347), "cop20k": (S.GEN_COP20K_LIKE, 0, 0, 12346)}
m = S.generate(*W[name])
eng = S.Engine(); d = eng.upload(m)
cap = S.count_products(m, m, 0, m.shape[0])
out = []
for _ in range(14):
    eng.fused_owned(d, d, 0, m.shape[0], cap)
    out.append(round(eng.stats()["ms_task"] * 1e3))
print("SPADA_SHOP", os.environ.get("SPADA_SHOP"), name, "k_task us per call:", out, flush=True)
eng.free(d); eng.close()
