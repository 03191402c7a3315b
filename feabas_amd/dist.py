"""Sharding and the (only) exchange steps of the multi-GPU runs.

One process per GPU (``torch.distributed``; backend "nccl" = RCCL over xGMI on the GPU box, "gloo"
in the CPU tests).  Tile pairs / sections are independent units: every rank works on a contiguous
shard -- the same contiguous-slice partitioning the reference uses for its worker jobs
(feabas/stitcher.py:375-392) -- with no collective on the data path.  What is exchanged afterwards:
  * the variable-length match table of every rank   -> ``gather_match_table``
  * per-section node displacement vectors           -> ``allgather_ragged``
"""
import numpy as np


def shard_range(n_items, rank, world):
    """[start, stop) of the contiguous shard of `rank`; shard sizes differ by at most one."""
    base, rem = divmod(int(n_items), int(world))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def _dist():
    import torch
    import torch.distributed as dist
    return torch, dist


def _device(dist):
    import torch
    return torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' else torch.device('cpu')


def allgather_ragged(arr, group=None):
    """all-gather of per-rank arrays whose first dimension differs.  Returns the list of every rank's array.
    Two collectives: the row counts, then one padded all_gather_into_tensor."""
    torch, dist = _dist()
    arr = np.ascontiguousarray(arr)
    world = dist.get_world_size(group)
    dev = _device(dist)
    cnt = torch.tensor([arr.shape[0]], dtype=torch.int64, device=dev)
    cnts = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(cnts, cnt, group=group)
    cnts = cnts.cpu().numpy()
    mx = int(cnts.max())
    tail = arr.shape[1:]
    pad = np.zeros((mx,) + tail, dtype=arr.dtype)
    pad[:arr.shape[0]] = arr
    t = torch.from_numpy(pad).to(dev)
    out = torch.empty((world * mx,) + tail, dtype=t.dtype, device=dev)
    dist.all_gather_into_tensor(out, t, group=group)
    out = out.cpu().numpy().reshape((world, mx) + tail)
    return [out[r, :cnts[r]] for r in range(world)]


def gather_match_table(pair_ids, xy0, xy1, weight, pair_offset=0, group=None):
    """Match tables of all ranks as one table [M, 6] = (global pair id, x0, y0, x1, y1, weight), float64,
    ordered by rank (= by global pair id for contiguous shards).  The on-disk layout of the reference
    concatenates xy0, xy1, weight per pair the same way (stitcher.py:144-151)."""
    tab = np.concatenate((np.asarray(pair_ids, dtype=np.float64).reshape(-1, 1) + pair_offset,
                          np.asarray(xy0, dtype=np.float64).reshape(-1, 2), np.asarray(xy1, dtype=np.float64).reshape(-1, 2),
                          np.asarray(weight, dtype=np.float64).reshape(-1, 1)), axis=1)
    parts = allgather_ragged(tab, group=group)
    return np.concatenate(parts, axis=0)
