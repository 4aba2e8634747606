#!/usr/bin/env python3
"""The expand-only floor under the task kernel (include/spada_probe.h, spgemm_probe.hip.hpp): static task assignment, descriptors and
entry records prefetched a task ahead, no ticket, no chain, no accumulator -- (i) products discarded, (ii) 12 bytes per product stored.
GPU box:  python scripts/probe_floor.py [webbase cop20k ...]   ->  the lines of profiles/r06_floor.txt"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spada_sim_amd as S
from spada_sim_amd import _ffi

W = {"webbase": (S.GEN_WEBBASE_LIKE, 0, 0, 12347), "cop20k": (S.GEN_COP20K_LIKE, 0, 0, 12346),
     "cage12": (S.GEN_CAGE12_LIKE, 0, 0, 12348), "mc2depi": (S.GEN_MC2DEPI_LIKE, 0, 0, 12349),
     "rmat16": (S.GEN_RMAT, 16, 16, 22)}
L = _ffi.lib()
fn = L.spada_dev_probe_floor
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_double),
               ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
names = sys.argv[1:] or ["webbase", "cop20k"]
eng = S.Engine()
for name in names:
    kind, p0, p1, seed = W[name]
    m = S.generate(kind, p0, p1, seed)
    d = eng.upload(m)
    cap = S.count_products(m, m, 0, m.shape[0])
    best = None
    for it in range(4):
        eng.fused_owned(d, d, 0, m.shape[0], cap)
        st = eng.stats()
        if best is None or st["ms_task"] < best["ms_task"]:
            best = st
    print(f"== {name}: products {best['nprod']} nnz(C) {best['c_nnz']} tasks {best['n_tasks']}; one pass {best['ms_fused_call']:.3f} ms, task kernel {best['ms_task']:.3f} ms")
    for write in (0, 1):
        for wgs in (4, 3, 2, 1):
            b, mean, nt, sk = ctypes.c_double(), ctypes.c_double(), ctypes.c_uint64(), ctypes.c_uint64()
            rc = fn(eng._ctx, write, wgs, 10, ctypes.byref(b), ctypes.byref(mean), ctypes.byref(nt), ctypes.byref(sk))
            if rc:
                print("   probe failed:", L.spada_last_error().decode())
                continue
            byt = best["nprod"] * 12 + best["a_nnz"] * 20
            print(f"   floor {'(ii) 12 B per product stored' if write else '(i) products discarded    '}  {wgs} workgroups / CU: best {b.value * 1e3:7.1f} us  mean {mean.value * 1e3:7.1f} us"
                  f"   ({byt / b.value / 1e6:6.0f} GB/s of gathered bytes; {sk.value} of {nt.value} tasks skipped)")
    eng.free(d)
