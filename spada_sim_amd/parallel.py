"""Multi-GPU plumbing: product-balanced A-row blocks (one per rank, B replicated) and the allgatherv
that concatenates the C row blocks on every rank.

The reference has no collective; its scheduler hands out disjoint A-row blocks
(scheduler.rs:296-379), which is what the ranks take here.  The only exchange step of the path is the
concatenation of C: counts first, then one variable-size all-gather per array.  With the `nccl` backend
(= RCCL on ROCm) torch runs an uneven all_gather as one group of per-rank broadcasts, i.e. every rank's
segment crosses each xGMI link once; `gloo` (CPU tests) gets the same result from explicit broadcasts.
torch.distributed is plumbing here: tensors wrap the device buffers the engine filled.
"""
import torch
import torch.distributed as dist


def _allgather_uneven(local, counts, group=None):
    """local: 1-D tensor of this rank; counts: python ints per rank.  Returns the concatenation."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    total = sum(counts)
    if dist.get_backend(group) != "nccl" and local.is_cuda:
        # gloo with device tensors (validation runs of the N > 1 path on a single GPU): stage through the host
        return _allgather_uneven(local.cpu(), counts, group).to(local.device)
    full = torch.empty(total, dtype=local.dtype, device=local.device)
    views, off = [], 0
    for n in counts:
        views.append(full[off:off + n])
        off += n
    if dist.get_backend(group) == "nccl":
        try:
            dist.all_gather(views, local.contiguous(), group=group)
        except (RuntimeError, ValueError):
            # backend without uneven all_gather: one equal-size RCCL all-gather of padded segments, then compaction
            cap = max(max(counts), 1)
            send = torch.zeros(cap, dtype=local.dtype, device=local.device)
            send[:local.numel()].copy_(local)
            recv = torch.empty(world * cap, dtype=local.dtype, device=local.device)
            dist.all_gather_into_tensor(recv, send, group=group)
            for r, n in enumerate(counts):
                views[r].copy_(recv[r * cap:r * cap + n])
    else:
        views[rank].copy_(local)
        works = [dist.broadcast(views[r], src=dist.get_global_rank(group, r) if group is not None else r,
                                group=group, async_op=True) for r in range(world) if counts[r] > 0]
        for w in works:
            w.wait()
    return full


def allgatherv_c(c_ptr, c_idx, c_val, group=None):
    """Concatenate per-rank C row blocks (local indptr starting at 0, indices, data) on every rank.
    Returns (indptr[int64, total_rows + 1], indices, data) of the whole C."""
    world = dist.get_world_size(group)
    dev = c_ptr.device
    cdev = dev if dist.get_backend(group) == "nccl" else torch.device("cpu")
    mine = torch.tensor([c_ptr.numel() - 1, c_idx.numel()], dtype=torch.int64, device=cdev)
    allc = torch.empty(world * 2, dtype=torch.int64, device=cdev)
    dist.all_gather_into_tensor(allc, mine, group=group)
    allc = allc.view(world, 2).cpu()
    rows = [int(x) for x in allc[:, 0]]
    nnzs = [int(x) for x in allc[:, 1]]
    idx = _allgather_uneven(c_idx, nnzs, group)
    val = _allgather_uneven(c_val, nnzs, group)
    lens = _allgather_uneven((c_ptr[1:] - c_ptr[:-1]).contiguous(), rows, group)
    indptr = torch.zeros(sum(rows) + 1, dtype=torch.int64, device=dev)
    torch.cumsum(lens, 0, out=indptr[1:])
    return indptr, idx, val
