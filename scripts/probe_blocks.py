import sys, os
sys.path.insert(0, os.getcwd())
import spada_sim_amd as S
m = S.generate(S.GEN_WEBBASE_LIKE, 0, 0, 12347)
eng = S.Engine(); d = eng.upload(m)
n = m.shape[0]
b = S.partition_rows(m, m, 8)
for (r0, r1) in [(0, n), (0, int(b[4])), (int(b[4]), n), (int(b[3]), int(b[4])), (int(b[7]), n)]:
    cap = S.count_products(m, m, r0, r1)
    best = None
    for it in range(6):
        eng.fused_owned(d, d, r0, r1, cap)
        st = eng.stats()
        if best is None or st["ms_fused_call"] < best["ms_fused_call"]: best = st
    st = best
    print(f"rows [{r0},{r1}) nprod {st['nprod']}: device {st['ms_fused_call']:.3f} stats {st['ms_row_stats']:.3f} big {st['ms_big_expand']:.3f} cut {st['ms_cut']:.3f} task {st['ms_task']:.3f}")
