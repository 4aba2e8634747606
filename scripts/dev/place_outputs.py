"""development aid (GPU box): does the one-pass task kernel's time depend on WHERE the caller's C buffers lie?  The web input, eight (indices, values) buffer
pairs allocated one after the other and all kept; k_task (us, best of 8 calls after 12 warm-up calls) into each pair, two passes over the pairs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import spada_sim_amd as S
name = sys.argv[1] if len(sys.argv) > 1 else "webbase"
W = {"webbase": (S.GEN_WEBBASE_LIKE, 0, 0, 12347), "cop20k": (S.GEN_COP20K_LIKE, 0, 0, 12346)}
m = S.generate(*W[name])
eng = S.Engine(); d = eng.upload(m)
rows = m.shape[0]
cap = S.count_products(m, m, 0, rows)
dev = torch.device("cuda", 0)
ptr = torch.empty(rows + 1, dtype=torch.int64, device=dev)
pairs = [(torch.empty(cap, dtype=torch.int32, device=dev), torch.empty(cap, dtype=torch.float64, device=dev)) for _ in range(8)]
for _ in range(12):
    eng.fused(d, d, 0, rows, ptr.data_ptr(), pairs[0][0].data_ptr(), pairs[0][1].data_ptr(), cap)
for rnd in range(2):
    out = []
    for idx, val in pairs:
        best = 1e9
        for _ in range(8):
            eng.fused(d, d, 0, rows, ptr.data_ptr(), idx.data_ptr(), val.data_ptr(), cap)
            best = min(best, eng.stats()["ms_task"] * 1e3)
        out.append(round(best, 1))
    print(name, "k_task us into each of eight buffer pairs:", out, flush=True)
eng.free(d); eng.close()
