import os, sys, json
sys.path.insert(0, os.getcwd())
import spada_sim_amd as S
sys.path.insert(0, "scripts")
from perf_probe import W
eng = S.Engine()
for name in sys.argv[1:]:
    kind, p0, p1, seed = W[name]
    m = S.generate(kind, p0, p1, seed)
    d = eng.upload(m)
    best = None
    for it in range(8):
        eng.symbolic(d, d, 0, m.shape[0]); eng.numeric_owned()
        st = eng.stats()
        if best is None or st["ms_num_flat"] < best["ms_num_flat"]: best = st
    print(f"{name}: num_flat {best['ms_num_flat']:.3f} sym_flat {best['ms_sym_flat']:.3f} numeric {best['ms_numeric_call']:.3f} symbolic {best['ms_symbolic_call']:.3f}")
    eng.free(d)
