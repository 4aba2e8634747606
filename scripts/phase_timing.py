#!/usr/bin/env python3
"""Per-phase shader-clock cycles of k_num_hash<G,*> on sampled workgroups (development aid, GPU box).
usage: SPADA_DBG_G=<G> phase_timing.py <workload>"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
os.environ.setdefault("SPADA_SERIAL_BINS", "1")
import spada_sim_amd as S
from spada_sim_amd import _ffi
from perf_probe import W
name = sys.argv[1] if len(sys.argv) > 1 else "webbase"
kind, p0, p1, seed = W[name]
m = S.generate(kind, p0, p1, seed)
eng = S.Engine()
d = eng.upload(m)
for it in range(3):
    eng.symbolic(d, d, 0, m.shape[0]); eng.numeric_owned()
L = _ffi.lib()
buf = np.zeros(64 * 16, np.uint64)
L.spada_debug_read.restype = ctypes.c_int
L.spada_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
rc = L.spada_debug_read(eng._ctx, buf.ctypes.data_as(ctypes.c_void_p), 64 * 16)
flat = os.environ.get("SPADA_DBG_G") in ("1", "2")
symk = os.environ.get("SPADA_DBG_G") == "2"
NS = 4 if symk else (7 if flat else 8)
t = buf.reshape(64, 16)[:, :NS].astype(np.int64)
ok = t[:, NS - 1] > t[:, 0]
t = t[ok]
names = ["rows+init", "walk", "zero bcnt", "bucket count", "scan", "scatter", "rank+write"] if not flat else ["rows+init", "walk", "bucket count", "scan", "scatter", "rank+write"]
if not flat: names = ["init", "walk", "keys+minmax", "bucket count", "scan", "scatter", "rank+write"]
if symk: names = ["rows+init", "walk", "write counts"]
dt = np.diff(t, axis=1)
print(f"G={os.environ.get('SPADA_DBG_G')} sampled {len(t)} workgroups; cycles (mean / median / max)")
for i, n in enumerate(names):
    print(f"  {n:14s} {dt[:, i].mean():9.0f} {np.median(dt[:, i]):9.0f} {dt[:, i].max():9.0f}")
print(f"  {'total':14s} {(t[:,NS-1]-t[:,0]).mean():9.0f}")

if flat:
    w = buf.reshape(64, 16)[:, 8:14].astype(np.int64)[ok]
    for i, n in enumerate(["entry search+issue", "entry wait+scan", "lds write+barrier", "search+gather issue", "hash", "final barrier"]):
        print(f"    walk/{n:22s} {w[:, i].mean():9.0f}")
