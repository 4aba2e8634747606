"""CPU, world_size 2, 3 and 8 over gloo: the N > 1 path of bench.py -- product-balanced A-row blocks
(one per rank, B replicated) and the allgatherv that concatenates the C row blocks on every rank
(spada_sim_amd/parallel.py).  On the GPU box the per-rank block is computed by the HIP engine; here,
without a GPU, each rank's block is produced by the oracle so that the exchange logic is what is tested."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ragged, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import spada_sim_amd as S
        from spada_sim_amd import parallel
        from oracle import oracle
        m = S.generate(S.GEN_RMAT, 10, 8, 17)
        if ragged and world == 8:   # two ranks end up with an empty row block (the first and one in the middle)
            q8 = m.shape[0] // 6
            bounds = [0, 0, q8, 2 * q8, 2 * q8, 3 * q8, 4 * q8, 5 * q8, m.shape[0]]
        elif ragged:   # one rank ends up with an empty row block
            bounds = [0, 0, m.shape[0]] if world == 2 else [0, 0, m.shape[0] // 2, m.shape[0]]
        else:
            bounds = S.partition_rows(m, m, world)
        r0, r1 = bounds[rank], bounds[rank + 1]
        ao = oracle.Csr(m.shape[0], m.shape[1], m.indptr, m.indices, m.data)
        sub = oracle.Csr(r1 - r0, m.shape[1], m.indptr[r0:r1 + 1] - m.indptr[r0],
                         m.indices[int(m.indptr[r0]):int(m.indptr[r1])], m.data[int(m.indptr[r0]):int(m.indptr[r1])])
        c = oracle.spgemm_spa(sub, ao, n_threads=1)
        c_ptr = torch.from_numpy(c.indptr.astype(np.int64))
        c_idx = torch.from_numpy(c.indices.astype(np.int32))
        c_val = torch.from_numpy(c.data.copy())
        indptr, idx, val = parallel.allgatherv_c(c_ptr, c_idx, c_val)
        full = oracle.spgemm_spa(ao, ao, n_threads=1)
        ok = (np.array_equal(indptr.numpy().astype(np.uint64), full.indptr)
              and np.array_equal(idx.numpy().astype(np.uint64), full.indices)
              and np.array_equal(val.numpy(), full.data))
        q.put((rank, bool(ok), int(indptr[-1])))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,ragged", [(2, False), (3, False), (2, True), (8, True)])
def test_row_blocks_allgatherv_reassembles_c(world, ragged):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ragged, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert len({n for _, _, n in res}) == 1     # every rank holds the same, complete C
