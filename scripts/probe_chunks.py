#!/usr/bin/env python3
"""Per-chunk statistics of the streamed product (development aid, GPU box): scripts/probe_chunks.py SCALE NCHUNKS chunk ..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import spada_sim_amd as S

scale, nchunks = int(sys.argv[1]), int(sys.argv[2])
which = [int(x) for x in sys.argv[3:]] or [0, nchunks // 2, nchunks - 1]
t0 = time.perf_counter()
m = S.generate(S.GEN_RMAT, scale, 16, 22)
print(f"generated in {time.perf_counter() - t0:.1f} s: rows {m.shape[0]} nnz {m.nnz()}", flush=True)
bounds = S.partition_rows(m, m, nchunks)
eng = S.Engine()
d = eng.upload(m)
dev = torch.device("cuda", 0)
keys = ("nprod", "c_nnz", "n_tasks", "multi_pass_tasks", "pipeline_runs", "spill_rows", "scratch_products", "task_product_limit",
        "ms_symbolic_call", "ms_numeric_call", "ms_row_stats", "ms_big_expand", "ms_cut", "ms_task", "cls_rows", "cls_prod")
for c in which:
    b0, b1 = int(bounds[c]), int(bounds[c + 1])
    for rep in range(2):
        tw = time.perf_counter()
        nnz = eng.symbolic(d, d, b0, b1)
        s1 = eng.stats()
        p = torch.empty(b1 - b0 + 1, dtype=torch.int64, device=dev)
        i = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
        v = torch.empty(max(nnz, 1), dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        ta = time.perf_counter()
        eng.numeric(p.data_ptr(), i.data_ptr(), v.data_ptr())
        s2 = eng.stats()
        wall = time.perf_counter() - tw
        del p, i, v
    print(f"== chunk {c}: rows [{b0}, {b1})  wall {wall * 1e3:.1f} ms (numeric call wall {(time.perf_counter() - ta) * 1e3:.1f})")
    print("   symbolic:", {k: s1[k] for k in keys if k in s1})
    print("   numeric :", {k: s2[k] for k in keys if k in s2}, flush=True)
eng.free(d)
