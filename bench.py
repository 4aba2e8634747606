#!/usr/bin/env python3
"""bench.py -- nnz(C)/s of A*A SpGEMM on MI355X, with roofline and CPU baseline (contract: see DESIGN.md).

One "step" = one complete SpGEMM of the workload: the one-pass task pipeline (row statistics, BIG-row stage, task list,
task kernel) into CALLER-OWNED C buffers that are allocated once, sized by the product count of the row block, and
reused from step to step (`"c_alloc": "reused"` in the JSON line; `--two-phase` times the symbolic + numeric contract
instead, whose buffers can only be sized after the symbolic call and are allocated in every step: `"c_alloc": "per_step"`),
and (for N > 1) the allgatherv of the C row blocks.  Inputs (A, B = A) are resident in HBM before the timed region
starts; the timed region is bracketed by a barrier + torch.cuda.synchronize() on both sides and the max over ranks is
taken.  `value` and `ms_per_step` are the mean over the K steps (the contract); `ms_per_step_median` is printed beside it.
Once, outside the timed region, the C of the timed entry point is checked (`"verified"`): against the other entry point
(structure identical, values within 1e-9) and, unless --no-cpu-baseline, against the CPU oracle's product of the same matrix.

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


def load_traffic(workload):
    """HBM bytes per launch of every kernel from the committed PMC passes (profiles/rNN_traffic_<workload>.json, written by
    scripts/collect_traffic.sh on the GPU box); (None, None) when that file is absent."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_traffic_{workload}.json")))   # the latest round's file
    if not found:
        return None, None
    path = found[-1]
    with open(path) as f:
        d = json.load(f)
    return d, path


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="webbase-1M")
    ap.add_argument("--accumulator", default="lds_hash", choices=["lds_hash", "sort_merge"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--two-phase", action="store_true", help="time spada_dev_spgemm_symbolic + _numeric instead of the one-pass entry point")
    ap.add_argument("--exchange", default="overlap", choices=["overlap", "after", "torch"],
                    help="N > 1: how the C row blocks are replicated -- overlap: libspada_comm.so, two-phase, every finished piece "
                         "of the own block is broadcast (RCCL) while the next is computed; after: one-pass SpGEMM of the block, then "
                         "the RCCL allgatherv of libspada_comm.so; torch: the same allgatherv through torch.distributed")
    ap.add_argument("--exchange-chunks", type=int, default=4)
    ap.add_argument("--exercise-exchange", action="store_true",
                    help="N = 1 only: run both libspada_comm.so exchange forms on a one-rank RCCL communicator, check them against the "
                         "plain one-pass result and exit (the code path the driver's multi-GPU runs take, on the one GPU a builder has)")
    ap.add_argument("--chunk-one-pass", action="store_true",
                    help="chunked steps (R-MAT 22) through spada_dev_spgemm_fused with reused buffers instead of symbolic + numeric")
    ap.add_argument("--chunk-consumer", choices=("none", "checksum"), default="none",
                    help="chunked steps: what reads a finished chunk of C before it is dropped -- nothing (default: the engine's calls alone, "
                         "as for the workloads whose C fits: nobody reads C there either) or a torch reduction over its values inside the "
                         "timed step (rounds 2 - 4 and profiles/r05_bench_rmat22_1gpu_checksum.json: 8 bytes per output read once more, "
                         "0.10 s of R-MAT 22's step)")
    ap.add_argument("--chunk-products", type=float, default=0,
                    help="stream C in A-row chunks of about this many products (0 = automatic: chunk when the product "
                         "count of a rank exceeds 3e9, i.e. when C would not fit next to the inputs)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import spada_sim_amd as S
    from spada_sim_amd._ffi import Stats as _ffi_stats
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
    # dropped (or first read by a checksum: --chunk-consumer) one after the other; nothing is gathered (C of the whole job would not fit one GPU)
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
    one_pass = not args.two_phase and args.accumulator == "lds_hash"
    # (one-pass mode: the capacity of a chunk's C buffers = its product count, known to the host before anything runs)
    chunk_caps = ([S.count_products(a, a, chunk_bounds[i], chunk_bounds[i + 1]) for i in range(len(chunk_bounds) - 1)]
                  if chunk_bounds is not None and one_pass and args.chunk_one_pass else None)
    if chunk_bounds is not None and chunk_caps is None:
        # chunked steps run the two-phase contract: measured on R-MAT 22, the one-pass call is TWICE as slow there (7.2 s against
        # 3.5 s per step: hub rows are thousands of range tasks, some of them multi-pass, and every task behind them in the
        # chain waits for their counts); --chunk-one-pass times it anyway
        one_pass = False
    cap = my_products      # capacity of the C buffers of the one-pass entry point: one entry per product at most
    checksum = torch.zeros(1, dtype=torch.float64, device=dev)
    gather_s = [0.0]   # seconds spent in the allgatherv of C (N > 1, exchange after the compute), timed steps only
    exchange = args.exchange if world > 1 and chunk_bounds is None else None
    if exchange in ("overlap", "after") and (backend != "nccl" or args.accumulator != "lds_hash" and exchange == "after"):
        exchange = "torch"     # several ranks per GPU (gloo validation runs): RCCL refuses that
    comm = None
    exchange_fallback = None      # why the native exchange was given up for the torch.distributed one, if it was
    if exchange in ("overlap", "after"):
        # the 128-byte RCCL id travels from rank 0 through the process group that launched us
        uid = [S.Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        try:
            comm = S.Comm(uid[0], rank, world, local_rank)
        except Exception as e:     # (every rank votes below: the exchange must be the same one everywhere)
            exchange_fallback = f"spada_comm_create failed on rank {rank}: {e}"

    def exchange_native(mode):
        """One step at N > 1 through libspada_comm.so; returns (stats, local nnz, (indptr, indices, data) of the whole C)."""
        if mode == "overlap":
            rows_r, nnz_r = comm.dist_symbolic(eng, da, da, r0, r1, args.exchange_chunks)
            st = eng.stats()
            total, trows = int(nnz_r.sum()), int(rows_r.sum())
            f_ptr = torch.empty(trows + 1, dtype=torch.int64, device=dev)
            fbuf = torch.empty(max(total, 1) * 12, dtype=torch.uint8, device=dev)
            f_val = fbuf[:max(total, 1) * 8].view(torch.float64)
            f_idx = fbuf[max(total, 1) * 8:].view(torch.int32)
            torch.cuda.synchronize()
            comm.dist_numeric(eng, f_ptr.data_ptr(), f_idx.data_ptr(), f_val.data_ptr())
            st.update({k: v for k, v in eng.stats().items() if k in ("ms_numeric_call",)})
            return st, int(nnz_r[rank]), (f_ptr, f_idx[:total], f_val[:total])
        c_ptr = torch.empty(r1 - r0 + 1, dtype=torch.int64, device=dev)
        buf = torch.empty(max(cap, 1) * 12, dtype=torch.uint8, device=dev)
        c_val = buf[:max(cap, 1) * 8].view(torch.float64)
        c_idx = buf[max(cap, 1) * 8:].view(torch.int32)
        torch.cuda.synchronize()
        nnz = eng.fused(da, da, r0, r1, c_ptr.data_ptr(), c_idx.data_ptr(), c_val.data_ptr(), cap)
        st = eng.stats()
        tg = time.perf_counter()
        rows_r, nnz_r = comm.allgather_counts(r1 - r0, nnz)
        total, trows = int(nnz_r.sum()), int(rows_r.sum())
        f_ptr = torch.empty(trows + 1, dtype=torch.int64, device=dev)
        fbuf = torch.empty(max(total, 1) * 12, dtype=torch.uint8, device=dev)
        f_val = fbuf[:max(total, 1) * 8].view(torch.float64)
        f_idx = fbuf[max(total, 1) * 8:].view(torch.int32)
        torch.cuda.synchronize()
        comm.allgatherv_c(c_ptr.data_ptr(), c_idx.data_ptr(), c_val.data_ptr(), rows_r, nnz_r, f_ptr.data_ptr(), f_idx.data_ptr(),
                          f_val.data_ptr())
        gather_s[0] += time.perf_counter() - tg
        return st, nnz, (f_ptr, f_idx[:total], f_val[:total])

    out_bufs = {}   # one-pass mode: C.indptr and the (values | indices) buffer, allocated once

    def step():
        if chunk_bounds is not None:
            bufs = {}
            agg = {"c_nnz": 0, "nprod": 0, "bytes_read": 0, "bytes_write": 0}
            tms = {}

            def alloc(nrows, nnz):
                if chunk_caps is not None:
                    # one-pass mode: the buffers are sized once, for the largest chunk, and every chunk is written into them
                    if "p" not in out_bufs:
                        mrows = max(chunk_bounds[i + 1] - chunk_bounds[i] for i in range(len(chunk_bounds) - 1))
                        out_bufs["p"] = torch.empty(mrows + 1, dtype=torch.int64, device=dev)
                        out_bufs["i"] = torch.empty(max(max(chunk_caps), 1), dtype=torch.int32, device=dev)
                        out_bufs["v"] = torch.empty(max(max(chunk_caps), 1), dtype=torch.float64, device=dev)
                    bufs.update(out_bufs)
                    return bufs["p"].data_ptr(), bufs["i"].data_ptr(), bufs["v"].data_ptr()
                bufs["p"] = torch.empty(nrows + 1, dtype=torch.int64, device=dev)
                bufs["i"] = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
                bufs["v"] = torch.empty(max(nnz, 1), dtype=torch.float64, device=dev)
                return bufs["p"].data_ptr(), bufs["i"].data_ptr(), bufs["v"].data_ptr()

            def consume(b0, b1, nnz, st):
                if args.chunk_consumer == "checksum":
                    checksum.add_(bufs["v"][:nnz].sum())
                    # the sum runs on torch's stream, the next chunk is written by the engine's own (non-blocking) stream: it must
                    # have read the buffer before the buffer is dropped and its memory handed to the next chunk
                    # (letting the sum of chunk k run NEXT to the engine's work on chunk k + 1 instead was measured in round 5:
                    # 3.12 s per step of R-MAT 22 against 2.98 s -- the reduction takes the scatter's HBM bandwidth; the wait stays)
                    torch.cuda.current_stream().synchronize()
                for k in ("c_nnz", "nprod", "bytes_read", "bytes_write"):
                    agg[k] += st[k]
                for k, v in st.items():
                    if k.startswith("ms_"):
                        tms[k] = tms.get(k, 0.0) + v
                for k in ("cls_rows", "cls_prod"):
                    agg[k] = [x + y for x, y in zip(agg.get(k, [0] * len(st[k])), st[k])]
                agg["n_tasks"] = agg.get("n_tasks", 0) + st["n_tasks"]

            if chunk_caps is not None:
                nnz = eng.fused_row_chunks(da, da, chunk_bounds, chunk_caps, alloc, consume)
            else:
                nnz = eng.spgemm_row_chunks(da, da, chunk_bounds, alloc, consume)
            st = dict(agg)
            st.update(tms)
            return st, nnz, None
        if comm is not None:
            return exchange_native(exchange)
        if one_pass:
            # C's values and column indices: one allocation sized by the product count of the row block -- an upper bound of
            # nnz(C) the host knows before anything runs on the GPU (values first: 8-byte aligned); the buffers are the caller's
            # and are reused from step to step
            if "c_ptr" not in out_bufs:
                out_bufs["c_ptr"] = torch.empty(r1 - r0 + 1, dtype=torch.int64, device=dev)
                voff = (max(cap, 1) * 8 + 255) // 256 * 256       # (the index part starts on a 256-byte boundary, like the value part)
                out_bufs["buf"] = torch.empty(voff + max(cap, 1) * 4, dtype=torch.uint8, device=dev)
                # (the views and their device pointers are taken once as well: three tensor operations per step are ~10 us of
                # interpreter time that no caller of the C ABI spends)
                out_bufs["c_val"] = out_bufs["buf"][:max(cap, 1) * 8].view(torch.float64)
                out_bufs["c_idx"] = out_bufs["buf"][voff:].view(torch.int32)
                out_bufs["ptrs"] = (out_bufs["c_ptr"].data_ptr(), out_bufs["c_idx"].data_ptr(), out_bufs["c_val"].data_ptr())
            c_ptr, c_idx, c_val = out_bufs["c_ptr"], out_bufs["c_idx"], out_bufs["c_val"]
            nnz = eng.fused(da, da, r0, r1, *out_bufs["ptrs"], cap)
        else:
            c_ptr = torch.empty(r1 - r0 + 1, dtype=torch.int64, device=dev)
            nnz = eng.symbolic(da, da, r0, r1)
            voff = (max(nnz, 1) * 8 + 255) // 256 * 256
            buf = torch.empty(voff + max(nnz, 1) * 4, dtype=torch.uint8, device=dev)
            c_val = buf[:max(nnz, 1) * 8].view(torch.float64)
            c_idx = buf[voff:].view(torch.int32)
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

    if args.exercise_exchange:
        if world != 1:
            raise SystemExit("--exercise-exchange is a single-process check")
        comm = S.Comm(S.Comm.unique_id(), 0, 1, local_rank)
        comm_saved, comm = comm, None
        _, nnz, (p0, i0, v0) = step()
        comm = comm_saved
        for mode in ("overlap", "after"):
            _, n2, (fp, fi, fv) = exchange_native(mode)
            assert n2 == nnz and torch.equal(fp, p0) and torch.equal(fi, i0[:nnz]) and torch.allclose(fv, v0[:nnz], rtol=1e-9, atol=0), mode
        print(json.dumps({"exercise_exchange": "ok", "nnz_c": nnz, "modes": ["overlap", "after"]}))
        comm.close()
        eng.free(da)
        eng.close()
        return
    if exchange in ("overlap", "after"):
        # the native exchange against the torch.distributed one, once, before anything is timed: structure identical, values
        # within 1e-9; if a rank could not create its communicator, or the two disagree anywhere, EVERY rank falls back to the
        # torch.distributed exchange (and the JSON line says so)
        if comm is not None and exchange_fallback is None:
            try:
                _, _, (fp, fi, fv) = exchange_native(exchange)
                c_ptr = torch.empty(r1 - r0 + 1, dtype=torch.int64, device=dev)
                nnz = eng.symbolic(da, da, r0, r1)
                c_idx = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
                c_val = torch.empty(max(nnz, 1), dtype=torch.float64, device=dev)
                eng.numeric(c_ptr.data_ptr(), c_idx.data_ptr(), c_val.data_ptr())
                tp, ti, tv = parallel.allgatherv_c(c_ptr, c_idx[:nnz], c_val[:nnz])
                if not (fi.numel() == ti.numel() and bool(torch.equal(fp, tp)) and bool(torch.equal(fi, ti)) and
                        bool(torch.allclose(fv, tv, rtol=1e-9, atol=0))):
                    exchange_fallback = f"libspada_comm.so ({exchange}) and torch.distributed disagree on the gathered C (rank {rank})"
                del fp, fi, fv, tp, ti, tv
            except Exception as e:
                exchange_fallback = f"native exchange failed on rank {rank}: {e}"
        vote = torch.tensor([0 if exchange_fallback is None else 1], dtype=torch.int32, device=dev)
        dist.all_reduce(vote, op=dist.ReduceOp.MAX)
        if int(vote.item()):
            exchange_fallback = exchange_fallback or "another rank gave up the native exchange"
            if comm is not None:
                comm.close()
            comm, exchange = None, "torch"
    # The split of a call into its small kernels (row statistics / BIG-row stage / cut) needs an event record between them, and
    # each record idles the stream for about 5 us: the split is taken in the warm-up steps (one extra step if there are none)
    # and the records are switched off for the timed steps, where only the call and k_task are bracketed by events.
    PHASES = ("ms_row_stats", "ms_big_expand", "ms_cut")
    phase_acc, phase_n = {k: 0.0 for k in PHASES}, 0
    n_warm = max(args.warmup, 1)
    for w in range(n_warm):
        st_w, _, _ = step()
        if w == 0 and n_warm > 1:
            continue       # (the context's first call: workspaces grow and the row statistics are read back mid-run -- `first_call_ms` prices that)
        for k in PHASES:
            phase_acc[k] += st_w.get(k, 0.0)
        phase_n += 1
    eng.set_phase_timing(False)
    sync()
    compute_only_s = None
    if world > 1 and chunk_bounds is None:
        # the block computation alone (C left sharded), same number of steps: what the exchange is compared with
        sync()
        tc = time.perf_counter()
        for _ in range(args.steps):
            c_ptr = torch.empty(r1 - r0 + 1, dtype=torch.int64, device=dev)
            buf = torch.empty(max(cap, 1) * 12, dtype=torch.uint8, device=dev)
            if one_pass and exchange != "overlap":      # (the overlapped exchange times the two-phase contract: compare like with like)
                eng.fused(da, da, r0, r1, c_ptr.data_ptr(), buf[max(cap, 1) * 8:].data_ptr(), buf.data_ptr(), cap)
            else:
                nnz = eng.symbolic(da, da, r0, r1)
                eng.numeric(c_ptr.data_ptr(), buf[max(cap, 1) * 8:].data_ptr(), buf.data_ptr())
        sync()
        compute_only_s = time.perf_counter() - tc
    gather_s[0] = 0.0
    t0 = time.perf_counter()
    acc = {}
    TIMES = ("ms_fused_call", "ms_symbolic_call", "ms_numeric_call", "ms_row_stats", "ms_big_expand", "ms_cut", "ms_task")
    step_wall = []
    if world == 1 and one_pass and chunk_bounds is None and comm is None:
        # the plain one-pass step, as a caller of the C ABI would loop over it: the call and ONE read of its statistics (the C
        # structure itself -- the dictionary step() builds from it costs ~35 us of interpreter time per step, 3 % of a step that no
        # caller of the library spends; the engine's work inside the loop is the same)
        fused, ptrs, raw = eng.fused, out_bufs["ptrs"], _ffi_stats()
        acc = {k: 0.0 for k in TIMES}
        for _ in range(args.steps):
            ts = time.perf_counter()
            nnz_local = fused(da, da, r0, r1, *ptrs, cap)
            eng.stats_raw(raw)
            step_wall.append(time.perf_counter() - ts)     # (every step returns after its product is complete)
            for k in TIMES:
                acc[k] += getattr(raw, k)
        st = eng.stats()
    else:
        for _ in range(args.steps):
            ts = time.perf_counter()
            st, nnz_local, _ = step()
            step_wall.append(time.perf_counter() - ts)     # (every step returns after its stream has drained)
            for k in TIMES:
                acc[k] = acc.get(k, 0.0) + st.get(k, 0.0)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        rdev = dev if dist.get_backend() == "nccl" else torch.device("cpu")
        t = torch.tensor([elapsed, compute_only_s if compute_only_s is not None else elapsed - gather_s[0], gather_s[0]],
                         dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, compute_max_s, gather_max_s = (float(x) for x in t.tolist())
        tot = torch.tensor([st["c_nnz"], st["nprod"], st["bytes_read"], st["bytes_write"]], dtype=torch.int64, device=rdev)
        dist.all_reduce(tot)
        nnz_total, nprod_total = int(tot[0]), int(tot[1])
    else:
        nnz_total, nprod_total = st["c_nnz"], st["nprod"]

    # ---- verification, outside the timed region: the C of the entry point that was timed -----------------------------------
    verified = None
    if world == 1 and chunk_bounds is None:
        verified = {}
        try:
            _, nnz_t, (t_ptr, t_idx, t_val) = step()
            t_ptr, t_idx, t_val = t_ptr.clone(), t_idx[:nnz_t].clone(), t_val[:nnz_t].clone()
            if one_pass or args.accumulator != "lds_hash":
                # the other entry point: the symbolic + numeric contract (sort-merge runs: the one-pass call)
                if one_pass:
                    o_nnz = eng.symbolic(da, da, r0, r1)
                    o_ptr = torch.empty(r1 - r0 + 1, dtype=torch.int64, device=dev)
                    o_idx = torch.empty(max(o_nnz, 1), dtype=torch.int32, device=dev)
                    o_val = torch.empty(max(o_nnz, 1), dtype=torch.float64, device=dev)
                    eng.numeric(o_ptr.data_ptr(), o_idx.data_ptr(), o_val.data_ptr())
                    other = "spada_dev_spgemm_symbolic + _numeric"
                else:
                    o_ptr = torch.empty(r1 - r0 + 1, dtype=torch.int64, device=dev)
                    o_idx = torch.empty(max(cap, 1), dtype=torch.int32, device=dev)
                    o_val = torch.empty(max(cap, 1), dtype=torch.float64, device=dev)
                    o_nnz = eng.fused(da, da, r0, r1, o_ptr.data_ptr(), o_idx.data_ptr(), o_val.data_ptr(), cap)
                    other = "spada_dev_spgemm_fused"
                torch.cuda.synchronize()
                verified["other_entry_point"] = other
                verified["entry_points_agree"] = bool(
                    o_nnz == nnz_t and torch.equal(o_ptr, t_ptr) and torch.equal(o_idx[:o_nnz], t_idx) and
                    torch.all((o_val[:o_nnz] - t_val).abs() <= 1e-9 * t_val.abs()).item())
                del o_ptr, o_idx, o_val
            if not args.no_cpu_baseline and st["nprod"] <= 1.0e9:
                # the CPU oracle on the whole matrix (the checker, never the thing measured): structure bit-exact, values 1e-9
                from oracle import oracle
                ao = oracle.Csr(a.shape[0], a.shape[1], a.indptr, a.indices, a.data)
                ref = oracle.spgemm_spa(ao, ao)
                h_ptr = t_ptr.cpu().numpy().astype(np.uint64)
                h_idx = t_idx.cpu().numpy().astype(np.uint32).astype(np.uint64)
                h_val = t_val.cpu().numpy()
                verified["oracle_agrees"] = bool(
                    nnz_t == ref.nnz and np.array_equal(h_ptr, ref.indptr) and np.array_equal(h_idx, ref.indices) and
                    np.all(np.abs(h_val - ref.data) <= 1e-9 * np.abs(ref.data)))
                del ref, h_ptr, h_idx, h_val
            else:
                verified["oracle_agrees"] = None
            verified["nnz_c"] = int(nnz_t)
            verified["ok"] = all(v is not False for k, v in verified.items() if k.endswith("agree") or k.endswith("agrees"))
            del t_ptr, t_idx, t_val
        except Exception as e:      # a failed check must not lose the measurement: the line says what happened
            verified = {"ok": False, "error": f"{type(e).__name__}: {e}"}

    elif world == 1 and args.accumulator == "lds_hash":
        # chunked steps: the whole C is never resident; the first chunk through both entry points (the full-size parity tests of
        # this configuration: tests/test_gpu_tasks.py::test_rmat22_row_ranges_against_oracle)
        verified = {}
        try:
            b0, b1 = int(chunk_bounds[0]), int(chunk_bounds[1])
            cap0 = int(S.count_products(a, a, b0, b1))
            t_ptr = torch.empty(b1 - b0 + 1, dtype=torch.int64, device=dev)
            t_idx = torch.empty(max(cap0, 1), dtype=torch.int32, device=dev)
            t_val = torch.empty(max(cap0, 1), dtype=torch.float64, device=dev)
            nnz_t = eng.fused(da, da, b0, b1, t_ptr.data_ptr(), t_idx.data_ptr(), t_val.data_ptr(), cap0)
            o_nnz = eng.symbolic(da, da, b0, b1)
            o_ptr = torch.empty(b1 - b0 + 1, dtype=torch.int64, device=dev)
            o_idx = torch.empty(max(o_nnz, 1), dtype=torch.int32, device=dev)
            o_val = torch.empty(max(o_nnz, 1), dtype=torch.float64, device=dev)
            eng.numeric(o_ptr.data_ptr(), o_idx.data_ptr(), o_val.data_ptr())
            torch.cuda.synchronize()
            verified["other_entry_point"] = "spada_dev_spgemm_symbolic + _numeric (first chunk: rows %d .. %d)" % (b0, b1)
            verified["entry_points_agree"] = bool(
                o_nnz == nnz_t and torch.equal(o_ptr, t_ptr) and torch.equal(o_idx, t_idx[:nnz_t]) and
                torch.all((o_val - t_val[:nnz_t]).abs() <= 1e-9 * o_val.abs()).item())
            verified["oracle_agrees"] = None
            verified["nnz_c"] = int(nnz_t)
            verified["ok"] = verified["entry_points_agree"]
            del t_ptr, t_idx, t_val, o_ptr, o_idx, o_val
        except Exception as e:
            verified = {"ok": False, "error": f"{type(e).__name__}: {e}"}

    # ---- beside the headline, outside the timed region (N = 1, whole-matrix workloads): what ONE call on a fresh context costs, what
    # the two-phase contract of SURVEY 8(b) costs per step, and what the upload spent on the arrays it derives from the matrix ------------
    extra = {}
    if world == 1 and chunk_bounds is None and comm is None and args.accumulator == "lds_hash":
        import ctypes
        from spada_sim_amd import _ffi
        try:
            # (a) the reference does ONE execute() per process (main.rs:93): a fresh context, freshly uploaded operands, fresh C buffers --
            # workspaces grow inside this call, nothing is known about the input.  Wall time of the call; the runtime itself is warm.
            fc = []
            for rep in range(3):
                e2 = S.Engine(device=local_rank)
                d2 = e2.upload(a)
                p2 = torch.empty(rows + 1, dtype=torch.int64, device=dev)
                voff = (max(cap, 1) * 8 + 255) // 256 * 256
                b2 = torch.empty(voff + max(cap, 1) * 4, dtype=torch.uint8, device=dev)
                torch.cuda.synchronize()
                ts = time.perf_counter()
                e2.fused(d2, d2, 0, rows, p2.data_ptr(), b2[voff:].data_ptr(), b2.data_ptr(), cap)
                wall = (time.perf_counter() - ts) * 1e3
                s2 = e2.stats()
                fc.append({"ms_wall": wall, "ms_device_last_run": s2["ms_fused_call"], "pipeline_runs": s2["pipeline_runs"],
                           "pipeline_kind": "count + numeric" if s2["pipeline_kind"] else "one pass",
                           "workspace_mb": s2["workspace_bytes"] / 1e6})
                if rep == 0:
                    hm, dm, nb = ctypes.c_double(), ctypes.c_double(), ctypes.c_uint64()
                    fn = _ffi.lib().spada_dev_csr_aux_cost
                    fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                                             ctypes.POINTER(ctypes.c_uint64)]
                    if fn(d2, ctypes.byref(hm), ctypes.byref(dm), ctypes.byref(nb)) == 0:
                        extra["upload_aux_ms"] = {"host": hm.value, "device_row_ids_and_extents_kernels": dm.value, "bytes": nb.value,
                                                  "note": "rowid (4 B x nnz) and rext (8 B x rows) are derived on the device at upload, outside every "
                                                          "timed region; k_entry_stats reads them"}
                e2.free(d2)
                e2.close()
                del p2, b2
            extra["first_call"] = min(fc, key=lambda r: r["ms_wall"])
            extra["first_call"]["all_ms_wall"] = [r["ms_wall"] for r in fc]
            extra["first_call_ms"] = extra["first_call"]["ms_wall"]
        except Exception as e:
            extra["first_call"] = {"error": f"{type(e).__name__}: {e}"}
        try:
            # (b) the contract of SURVEY 8(b): symbolic -> the caller allocates C -> numeric, C allocated in every step (torch's caching allocator)
            KC = max(5, min(args.steps, 20))
            eng.set_phase_timing(False)
            for _ in range(2):
                n_ = eng.symbolic(da, da, r0, r1)
            torch.cuda.synchronize()
            tcs, dev_c = time.perf_counter(), 0.0
            for _ in range(KC):
                c_ptr = torch.empty(r1 - r0 + 1, dtype=torch.int64, device=dev)
                n_ = eng.symbolic(da, da, r0, r1)
                ms_sym = eng.stats_raw().ms_symbolic_call
                voff = (max(n_, 1) * 8 + 255) // 256 * 256
                buf = torch.empty(voff + max(n_, 1) * 4, dtype=torch.uint8, device=dev)
                eng.numeric(c_ptr.data_ptr(), buf[voff:].data_ptr(), buf.data_ptr())
                dev_c += ms_sym + eng.stats_raw().ms_numeric_call
            torch.cuda.synchronize()
            extra["contract_ms_per_step"] = (time.perf_counter() - tcs) / KC * 1e3
            extra["contract"] = {"entry_points": "spada_dev_spgemm_symbolic + spada_dev_spgemm_numeric, C allocated per step", "steps": KC,
                                 "device_ms_per_step": dev_c / KC}
        except Exception as e:
            extra["contract"] = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        K = args.steps
        ms_step = elapsed / K * 1e3
        ms = {k: v / K for k, v in acc.items()}
        for k in PHASES:
            ms[k] = phase_acc[k] / max(phase_n, 1)
        # device time of one SpGEMM on rank 0 (HIP events on the engine stream)
        dev_ms = ms["ms_fused_call"] if ms["ms_fused_call"] > 0 else ms["ms_symbolic_call"] + ms["ms_numeric_call"]
        pipe_gbs = st["bytes_read"] / (dev_ms * 1e-3) / 1e9
        # Dominant kernel: k_task -- the persistent task kernel that expands, scales, accumulates and orders every product of
        # the step (all rows; 60-70 % of the device time).  Its average duration comes from HIP events recorded around it on
        # the stream it is launched on; its algorithmic bytes are those of the whole step (12 B per product,
        # 28 B per A entry, 8 B per row read; 12 B per nnz(C) written), the other kernels are listed under `kernels`.
        k_ms = ms["ms_task"] if ms["ms_fused_call"] > 0 else dev_ms     # (chunked steps: summed over the chunks)
        achieved = st["bytes_read"] / (k_ms * 1e-3) / 1e9
        traffic, traffic_src = load_traffic(args.workload)
        cls_names = ["empty", "copy (one A entry)", "small", "solo", "big (column-range tasks)"]
        kernels = [
            {"kernel": "k_entry_stats + k_row_class_cut (B-row descriptors, products per row, row classes, the tiles cut into batches)", "ms": ms["ms_row_stats"]},
            {"kernel": "k_big_parts/hist/plan on the engine stream (big rows: parts, column histograms, ranges; the scatter of spilled rows and the cut table too when they are not forked to the side streams)",
             "ms": ms["ms_big_expand"], "products": st["cls_prod"][4] if "cls_prod" in st else None,
             "spilled_products": st.get("scratch_products"), "spilled_rows": st.get("spill_rows")},
            {"kernel": "from the plan's end to the task kernel's start: k_cut3 (task list) and, next to it on the side streams, k_big_scatter and k_big_cuts",
             "ms": ms["ms_cut"]},
        ]
        two_phase = not (ms["ms_fused_call"] > 0)
        # a two-phase step whose numeric phase was not timed (0 ms) would price the whole step at the symbolic time alone: no
        # roofline is printed from such a record
        roofline_valid = not two_phase or ms["ms_numeric_call"] > 0
        if two_phase:
            # (the phase times of a two-phase step: the counters of the symbolic call survive the numeric call except ms_task)
            kernels += [
                {"kernel": "k_task<COUNT> + k_pos1-4 (expand - accumulate keys, counts scanned into C.indptr; no chain)",
                 "ms": ms["ms_symbolic_call"] - ms["ms_row_stats"] - ms["ms_big_expand"] - ms["ms_cut"],
                 "products": nprod_total if world == 1 else st["nprod"], "tasks": st.get("n_tasks")},
                {"kernel": "k_task<NUMERIC> (expand - scale - accumulate - order at the known positions)", "ms": ms["ms_numeric_call"],
                 "products": nprod_total if world == 1 else st["nprod"], "tasks": st.get("n_tasks")}]
        else:
            inside_two = ms.get("ms_symbolic_call", 0.0) > 0   # the engine ran its count + numeric pipeline inside the one-pass entry point
            kernels += [{"kernel": "k_task<COUNT> + k_task<NUMERIC> (most products of this input lie in BIG rows: by its rule the engine runs count + numeric "
                                   "inside spada_dev_spgemm_fused: no chain)" if inside_two else
                                   "k_task (expand - scale - accumulate - order, all rows)", "ms": ms["ms_task"],
                         "products": nprod_total if world == 1 else st["nprod"], "tasks": st.get("n_tasks")}]
        out = {
            "metric": "nnz(C)/sec on A*A SpGEMM",
            "value": nnz_total / (elapsed / K),
            "unit": "nnz(C)/s",
            "n_gpus": world,
            "steps": K,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "ms_per_step_median": float(np.median(step_wall)) * 1e3,   # per-step wall times on rank 0 (every step drains its stream)
            "first_call_ms": extra.get("first_call_ms"),                # one call on a FRESH context and fresh operands (wall; details: first_call)
            "contract_ms_per_step": extra.get("contract_ms_per_step"),  # the two-phase contract of SURVEY 8(b), C allocated per step (wall)
            "first_call": extra.get("first_call"),
            "contract": extra.get("contract"),
            "upload_aux_ms": extra.get("upload_aux_ms"),
            "verified": None if verified is None else bool(verified.get("ok")),
            "verification": verified,
            "c_alloc": "reused" if one_pass and comm is None and (world == 1 or chunk_bounds is not None) else "per_step",
            # what one timed step consists of, as a key round-to-round comparisons can check (rounds 2 - 4 timed the chunked R-MAT 22 step
            # with the checksum of every chunk inside it; since round 5 the default leaves the chunks unread)
            "step_definition": ("chunked:" + ("one_pass" if one_pass else "two_phase") + ":consumer=" + args.chunk_consumer) if chunk_bounds is not None
                               else ("one_pass" if one_pass else "two_phase") + (":exchange=" + str(exchange) if world > 1 else ""),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": data_desc,
            "multi_gpu": None if world == 1 else {
                "exchange": {"overlap": "libspada_comm.so: two-phase, finished pieces of the own block broadcast (RCCL) while the next "
                                        f"is computed, {args.exchange_chunks} pieces", "after": "libspada_comm.so: RCCL allgatherv after the "
                                        "one-pass SpGEMM of the block", "torch": "torch.distributed allgatherv after the SpGEMM of the block",
                             None: "none (C streamed in row chunks, not gathered)"}[exchange],
                "exchange_fallback": exchange_fallback,
                "allgatherv_ms_per_step": gather_max_s / K * 1e3 if exchange != "overlap" else None,   # max over ranks
                "compute_ms_per_step": compute_max_s / K * 1e3,            # max over ranks: the block SpGEMM alone, timed separately
                "value_compute_only": nnz_total / (compute_max_s / K),     # nnz(C)/s with C left sharded by row block
                # what the replicated C costs at best: every GPU takes in (N - 1) / N of 12 B x nnz(C) over its seven xGMI links
                # (~100 GB/s each in practice) -- to be read next to compute_ms_per_step
                "predicted_ingest_ms": (world - 1) / world * 12.0 * nnz_total / (7 * 100e9) * 1e3,
                "note": "value includes the allgatherv that replicates C on every rank (north_star); every GPU must take in "
                        "(N-1)/N of 12 B x nnz(C) over xGMI, which bounds strong scaling once that exceeds the compute time"},
            "config": {"workload": f"{args.workload} A*A", "rows": rows, "nnz_a": a.nnz(), "products": nprod_total,
                       "nnz_c": nnz_total, "accumulator": args.accumulator,
                       "entry_point": ("spada_dist_spgemm_symbolic + spada_dist_spgemm_numeric (libspada_comm.so: two-phase, numeric phase in "
                                       f"{args.exchange_chunks} pieces overlapped with their broadcast)") if exchange == "overlap" else
                                      ("spada_dev_spgemm_fused (C buffers sized by the product count; the engine runs one pass or -- more than half of the "
                                       "products in BIG rows -- count + numeric inside the call: " +
                                       ("count + numeric" if ms.get("ms_symbolic_call", 0.0) > 0 and ms.get("ms_fused_call", 0.0) > 0 else "one pass") + ")")
                                      if one_pass
                                      else "spada_dev_spgemm_symbolic + spada_dev_spgemm_numeric",
                       # the headline runs on the additive one-pass entry point; the contract of SURVEY 8(b) is symbolic + numeric
                       # (`--two-phase` times it: profiles/r04_bench_webbase-1M_two_phase.json)
                       "entry_point_contract": ("additive one-pass entry point (spada_dev_spgemm_fused); the two-phase contract of SURVEY 8(b) "
                                                "-- spada_dev_spgemm_symbolic + _numeric -- is what --two-phase times") if one_pass and exchange != "overlap"
                                               else "two-phase contract of SURVEY 8(b): symbolic + numeric",
                       "parallelism": f"row-block x{world}, B replicated" +
                                      (f", C streamed in {len(chunk_bounds) - 1} row chunks per rank and not gathered; every finished chunk "
                                       + ("read once by a checksum of its values inside the timed step (--chunk-consumer)"
                                          if args.chunk_consumer == "checksum" else "dropped unread (--chunk-consumer none)")
                                       if chunk_bounds is not None else (", allgatherv of C" if world > 1 else ""))},
            "roofline": {
                "bound": "hbm",
                "kernel": ("k_task (persistent task kernel: expand - scale - accumulate in an ordered LDS block table - emission of every row of C); "
                           "average duration per step by HIP events on its stream, rank 0") if not two_phase else
                          ("whole device time of the symbolic + numeric calls (k_task<COUNT>, k_task<NUMERIC> and the kernels around "
                           "them) against the algorithmic bytes of ONE pass over the products; HIP events on the engine stream, rank 0"),
                "achieved": achieved if roofline_valid else None,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS if roofline_valid else None,
                "error": None if roofline_valid else "the numeric phase of the two-phase step reported 0 ms: not priced",
                "traffic": max([v.get("hbm_bytes_per_launch") or 0 for k, v in (traffic or {}).items()
                                if isinstance(v, dict) and k.startswith("k_task<2" if one_pass else "k_task<1")] or [None]) or None,
                "traffic_source": os.path.relpath(traffic_src, ROOT) if traffic else None,
                "kernel_ms": k_ms,
                "kernel_algorithmic_bytes_read": st["bytes_read"],
                "kernel_algorithmic_bytes_write": st["bytes_write"],
                "kernels": kernels,
                "row_classes": ({cls_names[k]: {"rows": st["cls_rows"][k], "products": st["cls_prod"][k]} for k in range(5)}
                                if "cls_rows" in st else None),
                "pipeline": {"achieved": pipe_gbs, "frac": pipe_gbs / HBM_PEAK_GBS, "device_ms_per_step": dev_ms,
                             "algorithmic_bytes_read": st["bytes_read"], "algorithmic_bytes_write": st["bytes_write"],
                             "phase_ms": ms,
                             "phase_ms_note": "ms_row_stats / ms_big_expand / ms_cut: average of the warm-up steps (their event "
                                              "records are switched off in the timed steps); the others: timed steps"},
            },
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(a)
        print(json.dumps(out))
    if comm is not None:
        comm.close()
    eng.free(da)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
