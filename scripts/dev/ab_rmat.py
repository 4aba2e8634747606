"""development aid (GPU box): wall time per one-pass call on R-MAT 18 / 16 for the library named by SPADA_LIB_PATH (same-box A/B of two builds)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import spada_sim_amd as S
for name, kind, p0, p1, seed in (("rmat18", S.GEN_RMAT, 18, 16, 22), ("rmat16", S.GEN_RMAT, 16, 16, 22)):
    m = S.generate(kind, p0, p1, seed)
    eng = S.Engine(); d = eng.upload(m)
    cap = S.count_products(m, m, 0, m.shape[0])
    eng.set_phase_timing(False)
    for _ in range(3):
        eng.fused_owned(d, d, 0, m.shape[0], cap)
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(10):
            eng.fused_owned(d, d, 0, m.shape[0], cap)
        best = min(best, (time.perf_counter() - t0) / 10 * 1e3)
    print(f"{name} {best:.4f} ms per call", flush=True)
    eng.free(d); eng.close()
