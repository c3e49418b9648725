"""Multi-GPU layer: one process per GPU, queries sharded contiguously by residue count, the packed reference
replicated, and ONE exchange step - an all-gather of the per-shard hit tables (fixed-size records + CIGAR arena)
before the order-dependent host pass / union-find.  torch.distributed is plumbing only: backend "nccl" is RCCL
over xGMI on the GPU box, "gloo" runs the same code on CPU tensors in the tests.

The reference has no counterpart (it is single-host multiprocessing with files as transport, uberBlast.py:333-338,
500-503); shards are independent because top-k is per query."""
import numpy as np


def shard_bounds(lengths, world):
    """contiguous query shards balanced by cumulative residue count; returns world+1 boundaries"""
    lengths = np.asarray(lengths, dtype=np.int64)
    n = len(lengths)
    if n == 0:
        return [0] * (world + 1)
    cum = np.concatenate([[0], np.cumsum(np.maximum(lengths, 1))])
    total = cum[-1]
    b = [0]
    for r in range(1, world):
        b.append(int(np.searchsorted(cum, total * r / world, side='left')))
    b.append(n)
    for i in range(1, len(b)):
        b[i] = max(b[i], b[i - 1])
    return b


def allgather_hits(hits, cigar, q_base, group=None, device=None):
    """hits: structured array (peppan_amd._native.HIT_DTYPE) with shard-local q indices; cigar: uint32 arena.
    Returns the concatenated (hits, cigar) of all ranks in rank order with global q indices and re-based
    cigar offsets.  Collectives: one all-gather of the two counts, one of a padded payload (hit records + arena)."""
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        if q_base:
            hits = hits.copy()
            hits['q'] += q_base
        return hits, cigar
    world = dist.get_world_size(group)
    dev = device if device is not None else torch.device('cpu')
    mine = hits.copy()
    mine['q'] += q_base
    counts = torch.tensor([len(mine), len(cigar)], dtype=torch.int64, device=dev)
    all_counts = torch.empty(2 * world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(all_counts, counts, group=group)
    all_counts = all_counts.cpu().numpy().reshape(world, 2)
    max_h, max_c = int(all_counts[:, 0].max()), int(all_counts[:, 1].max())
    rec = mine.dtype.itemsize
    # one payload per rank: [hit records, padded to max_h][CIGAR arena, padded to max_c] -> one collective, one copy back
    slot = max_h * rec + max_c * 4
    if slot == 0:
        return mine, np.zeros(0, dtype=np.uint32)
    buf = np.zeros(slot, dtype=np.uint8)
    buf[:len(mine) * rec] = mine.view(np.uint8).reshape(-1)
    buf[max_h * rec:max_h * rec + len(cigar) * 4] = np.ascontiguousarray(cigar, dtype=np.uint32).view(np.uint8)
    mine_t = torch.from_numpy(buf).to(dev)
    all_t = torch.empty(slot * world, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(all_t, mine_t, group=group)
    flat = all_t.cpu().numpy()
    out_h, out_c, coff = [], [], 0
    for r in range(world):
        nh, nc = int(all_counts[r, 0]), int(all_counts[r, 1])
        base = r * slot
        h = flat[base:base + nh * rec].view(mine.dtype).copy()
        h['cigar_off'] += coff
        out_h.append(h)
        out_c.append(flat[base + max_h * rec:base + max_h * rec + nc * 4].view(np.uint32))
        coff += nc
    return np.concatenate(out_h), np.concatenate(out_c)
