#!/usr/bin/env python3
"""bench.py -- nnz(C)/s of A*A SpGEMM on MI355X, with roofline and CPU baseline (contract: see DESIGN.md).

One "step" = one complete SpGEMM of the workload: symbolic phase, allocation of C, numeric phase and
(for N > 1) the allgatherv of the C row blocks.  Inputs (A, B = A) are resident in HBM before the
timed region starts; the timed region is bracketed by a barrier + torch.cuda.synchronize() on both sides
and the max over ranks is taken.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--accumulator lds_hash|sort_merge]

N > 1 is launched by the driver through torch.distributed.run (one rank per GPU, RCCL): the A rows are
split into product-balanced row blocks (one per rank), B is replicated, every rank computes its block
and the blocks are concatenated on every rank by an allgatherv over xGMI ("scaling": "strong" -- the
matrix is fixed as N grows).

Workloads: `webbase-1M` (BASELINE.json configs[2], the configuration the metric is quoted on), `cop20k_A`,
`cage12`, `mc2depi`, `rmat<scale>`.  The SuiteSparse files are not in the image and there is no network:
if $SPADA_MTX_DIR/<name>.mtx exists it is used, otherwise a seeded surrogate generator with matched
rows / nnz / degree profile is used and the JSON line says so ("data": "synthetic ...").
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 GB/s measured copy ceiling


def load_workload(S, name):
    gens = {
        "webbase-1M": (S.GEN_WEBBASE_LIKE, 0, 0, 12347),
        "cop20k_A": (S.GEN_COP20K_LIKE, 0, 0, 12346),
        "cage12": (S.GEN_CAGE12_LIKE, 0, 0, 12348),
        "mc2depi": (S.GEN_MC2DEPI_LIKE, 0, 0, 12349),
    }
    d = os.environ.get("SPADA_MTX_DIR")
    if d and os.path.exists(os.path.join(d, name + ".mtx")):
        return S.load_mm_mat(d, name), f"file {name}.mtx"
    if name.startswith("rmat"):
        scale = int(name[4:])
        return S.generate(S.GEN_RMAT, scale, 16, 22), f"synthetic R-MAT scale {scale} degree 16 seed 22"
    if name not in gens:
        raise SystemExit(f"unknown workload {name}")
    kind, p0, p1, seed = gens[name]
    return S.generate(kind, p0, p1, seed), f"synthetic surrogate of {name} (seeded generator, no SuiteSparse file in the image)"


def cpu_baseline(a, budget_s=12.0):
    """The oracle's OpenMP SPA variant (kind "port") on a bounded sample of the same workload: a row prefix sized by a
    2 % probe, repeated until about `budget_s` seconds of CPU work have been timed."""
    from oracle import oracle
    ao = oracle.Csr(a.shape[0], a.shape[1], a.indptr, a.indices, a.data)
    nt = oracle.num_threads()
    rows = a.shape[0]
    probe = max(1, rows // 50)

    def run(nrows):
        sub = oracle.Csr(nrows, a.shape[1], a.indptr[:nrows + 1], a.indices[:int(a.indptr[nrows])],
                         a.data[:int(a.indptr[nrows])])
        t0 = time.perf_counter()
        c = oracle.spgemm_spa(sub, ao, n_threads=nt)
        return time.perf_counter() - t0, c.nnz

    t, _ = run(probe)
    frac = min(1.0, budget_s / max(t, 1e-6) / 50.0)
    nrows = max(probe, int(rows * frac))
    total_t, total_nnz, reps = 0.0, 0, 0
    while total_t < budget_s and reps < 200:
        t, nnz = run(nrows)
        total_t += t
        total_nnz += nnz
        reps += 1
    return {"value": total_nnz / total_t, "unit": "nnz(C)/s", "cores": nt, "kind": "port",
            "sample": f"first {nrows} of {rows} A rows ({total_nnz // reps} nnz(C)) x {reps} repetitions in {total_t:.2f} s, "
                      f"oracle SPA variant, OpenMP {nt} threads"}


def pick_traffic(traffic, tmpl_suffix):
    """hbm_bytes_per_launch of the k_num_flat instantiation whose template list ends with `tmpl_suffix`."""
    if not traffic:
        return None
    for name, v in traffic.items():
        if name.startswith("k_num_flat<") and name.endswith(tmpl_suffix):
            return v["hbm_bytes_per_launch"]
    return None


def load_traffic(workload):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/r01_traffic_<workload>.json,
    written by scripts/collect_traffic.sh on the GPU box); None when that file is absent."""
    path = os.path.join(ROOT, "profiles", f"r01_traffic_{workload}.json")
    if not os.path.exists(path):
        return None, None
    with open(path) as f:
        d = json.load(f)
    return d, path


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="webbase-1M")
    ap.add_argument("--accumulator", default="lds_hash", choices=["lds_hash", "sort_merge"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--chunk-products", type=float, default=0,
                    help="stream C in A-row chunks of about this many products (0 = automatic: chunk when the product "
                         "count of a rank exceeds 3e9, i.e. when C would not fit next to the inputs)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import spada_sim_amd as S
    from spada_sim_amd import parallel

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch N > 1 through torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    # SPADA_BENCH_BACKEND=gloo: validation of the N > 1 path with several ranks sharing one GPU (RCCL refuses that);
    # the default and the measured configuration is nccl (= RCCL), one rank per GPU
    backend = os.environ.get("SPADA_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    a, data_desc = load_workload(S, args.workload)
    rows, cols = a.shape
    eng = S.Engine(device=local_rank, accumulator=S.ACC_SORT_MERGE if args.accumulator == "sort_merge" else S.ACC_LDS_HASH)
    bounds = S.partition_rows(a, a, world)
    r0, r1 = bounds[rank], bounds[rank + 1]
    da = eng.upload(a)                    # A and B = A resident in HBM before the timed region
    dev = torch.device("cuda", local_rank)

    # chunked mode (R-MAT scale 22): the rank's row block is cut into product-balanced row chunks whose C is produced,
    # checksummed and dropped one after the other; nothing is gathered (C of the whole job would not fit one GPU)
    my_products = S.count_products(a, a, r0, r1)
    chunk_products = args.chunk_products or (2.0 ** 31 if my_products > 3e9 else 0)
    chunk_bounds = None
    if chunk_products:
        nchunks = max(1, int(np.ceil(my_products / chunk_products)))
        fine = S.partition_rows(a, a, world * nchunks)
        chunk_bounds = [b for b in fine if r0 <= b <= r1]
        if chunk_bounds[0] != r0:
            chunk_bounds.insert(0, r0)
        if chunk_bounds[-1] != r1:
            chunk_bounds.append(r1)
    checksum = torch.zeros(1, dtype=torch.float64, device=dev)
    gather_s = [0.0]   # seconds spent in the allgatherv of C (N > 1), timed steps only

    def step():
        if chunk_bounds is not None:
            bufs = {}
            agg = {"c_nnz": 0, "nprod": 0, "bytes_read": 0, "bytes_write": 0}
            tms = {}

            def alloc(nrows, nnz):
                bufs["p"] = torch.empty(nrows + 1, dtype=torch.int64, device=dev)
                bufs["i"] = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
                bufs["v"] = torch.empty(max(nnz, 1), dtype=torch.float64, device=dev)
                return bufs["p"].data_ptr(), bufs["i"].data_ptr(), bufs["v"].data_ptr()

            def consume(b0, b1, nnz, st):
                checksum.add_(bufs["v"][:nnz].sum())
                for k in agg:
                    agg[k] += st[k]
                for k, v in st.items():
                    if k.startswith("ms_"):
                        tms[k] = tms.get(k, 0.0) + v
                for k in ("num_bin_rows", "num_bin_prod", "num_bin_entries", "num_bin_nnz"):
                    agg[k] = [x + y for x, y in zip(agg.get(k, [0] * len(st[k])), st[k])]

            nnz = eng.spgemm_row_chunks(da, da, chunk_bounds, alloc, consume)
            st = dict(agg)
            st.update(tms)
            return st, nnz, None
        c_ptr = torch.empty(r1 - r0 + 1, dtype=torch.int64, device=dev)       # size known before the symbolic phase
        nnz = eng.symbolic(da, da, r0, r1)
        # C's values and column indices: one allocation (values first: 8-byte aligned), sized by the symbolic result
        buf = torch.empty(max(nnz, 1) * 12, dtype=torch.uint8, device=dev)
        c_val = buf[:max(nnz, 1) * 8].view(torch.float64)
        c_idx = buf[max(nnz, 1) * 8:].view(torch.int32)
        eng.numeric(c_ptr.data_ptr(), c_idx.data_ptr(), c_val.data_ptr())     # returns after its stream has drained
        st = eng.stats()
        if world > 1:
            tg = time.perf_counter()
            full = parallel.allgatherv_c(c_ptr, c_idx[:nnz], c_val[:nnz])
            torch.cuda.synchronize()
            gather_s[0] += time.perf_counter() - tg
            return st, nnz, full
        return st, nnz, (c_ptr, c_idx, c_val)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    gather_s[0] = 0.0
    t0 = time.perf_counter()
    acc = {}
    for _ in range(args.steps):
        st, nnz_local, _ = step()
        for k in ("ms_symbolic_call", "ms_numeric_call", "ms_row_stats", "ms_binning", "ms_symbolic", "ms_scan",
                  "ms_numeric", "ms_sym_flat", "ms_num_flat", "ms_num_mid"):
            acc[k] = acc.get(k, 0.0) + st[k]
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        rdev = dev if dist.get_backend() == "nccl" else torch.device("cpu")
        t = torch.tensor([elapsed, elapsed - gather_s[0], gather_s[0]], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, compute_max_s, gather_max_s = (float(x) for x in t.tolist())
        tot = torch.tensor([st["c_nnz"], st["nprod"], st["bytes_read"], st["bytes_write"]], dtype=torch.int64, device=rdev)
        dist.all_reduce(tot)
        nnz_total, nprod_total = int(tot[0]), int(tot[1])
    else:
        nnz_total, nprod_total = st["c_nnz"], st["nprod"]

    if rank == 0:
        K = args.steps
        ms_step = elapsed / K * 1e3
        dev_ms = (acc["ms_symbolic_call"] + acc["ms_numeric_call"]) / K      # rank 0, HIP events on the engine stream
        pipe_gbs = st["bytes_read"] / (dev_ms * 1e-3) / 1e9
        # dominant kernel: the flat-batch numeric kernel k_num_flat, timed by HIP events on the stream it runs on.  It is
        # launched twice per step: shared batches of consecutive rows (bin 2) and list mode, one mid row per batch (bin 7);
        # the launch with the larger algorithmic byte count is reported as `roofline`, the other under `roofline.other`.
        def launch(bin_id, ms_key, label, tmpl):
            r_, p_, e_, n_ = (st[k][bin_id] for k in ("num_bin_rows", "num_bin_prod", "num_bin_entries", "num_bin_nnz"))
            rd, wr, ms = 12 * p_ + 28 * e_ + 8 * r_, 12 * n_ + 8 * r_, acc[ms_key] / K
            return {"label": label, "tmpl": tmpl, "ms": ms, "read": rd, "write": wr, "gbs": rd / (ms * 1e-3) / 1e9 if ms > 0 else 0.0,
                    "units": {"rows": r_, "products": p_, "a_entries": e_, "nnz_c": n_}}
        cands = [launch(2, "ms_num_flat", "shared batches of consecutive rows (nnz(C_i) <= 512)", "false, 1>"),
                 launch(7, "ms_num_mid", "list mode, one row per batch (768 < nnz(C_i) <= 1536)", "true, 1>")]
        if args.accumulator == "sort_merge":
            cands = cands[:1]
        cands.sort(key=lambda d: -d["read"])
        dom = cands[0]
        achieved, k_ms, k_read, k_write = dom["gbs"], dom["ms"], dom["read"], dom["write"]
        traffic, traffic_src = load_traffic(args.workload)
        out = {
            "metric": "nnz(C)/sec on A*A SpGEMM",
            "value": nnz_total / (elapsed / K),
            "unit": "nnz(C)/s",
            "n_gpus": world,
            "steps": K,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": data_desc,
            "multi_gpu": None if world == 1 else {
                "allgatherv_ms_per_step": gather_max_s / K * 1e3,          # max over ranks
                "compute_ms_per_step": compute_max_s / K * 1e3,            # max over ranks: everything but the gather
                "value_compute_only": nnz_total / (compute_max_s / K),     # nnz(C)/s with C left sharded by row block
                "note": "value includes the allgatherv that replicates C on every rank (north_star); every GPU must take in "
                        "(N-1)/N of 12 B x nnz(C) over xGMI, which bounds strong scaling once that exceeds the compute time"},
            "config": {"workload": f"{args.workload} A*A", "rows": rows, "nnz_a": a.nnz(), "products": nprod_total,
                       "nnz_c": nnz_total, "accumulator": args.accumulator,
                       "parallelism": f"row-block x{world}, B replicated" +
                                      (f", C streamed in {len(chunk_bounds) - 1} row chunks per rank and not gathered"
                                       if chunk_bounds is not None else (", allgatherv of C" if world > 1 else ""))},
            "roofline": {
                "bound": "hbm",
                "kernel": ("k_num_sortmerge (sort-merge accumulator over the rows with <= 1024 products)"
                           if args.accumulator == "sort_merge" else
                           "k_num_flat (flat-batch numeric: expand - scale - accumulate - order), " + dom["label"]) +
                          "; average duration per step by HIP events on its stream, rank 0",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pick_traffic(traffic, dom["tmpl"]) if args.accumulator != "sort_merge" else None,
                "traffic_source": os.path.relpath(traffic_src, ROOT) if traffic else None,
                "kernel_ms": k_ms,
                "kernel_units": dom["units"],
                "other": [{"kernel": "k_num_flat, " + o["label"], "kernel_ms": o["ms"], "achieved": o["gbs"],
                           "frac": o["gbs"] / HBM_PEAK_GBS, "kernel_units": o["units"]} for o in cands[1:]],
                "kernel_algorithmic_bytes_read": k_read,
                "kernel_algorithmic_bytes_write": k_write,
                "pipeline": {"achieved": pipe_gbs, "frac": pipe_gbs / HBM_PEAK_GBS, "device_ms_per_step": dev_ms,
                             "algorithmic_bytes_read": st["bytes_read"], "algorithmic_bytes_write": st["bytes_write"],
                             "phase_ms": {k: v / K for k, v in acc.items()}},
            },
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(a)
        print(json.dumps(out))
    eng.free(da)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
