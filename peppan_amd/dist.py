"""Multi-GPU layer of the all-vs-all search: one process per GPU, a 2-D grid of shards, ONE exchange step.

The ranks form an R x C grid (rank = r * C + c).  Row r owns a contiguous range of the QUERIES (balanced by residue count),
column c owns a contiguous range of the REFERENCE sequences - all frames and chunks of its genes, so K1, the streaming of the
targets through the seed index, the ungapped extensions and both Smith-Waterman passes all shrink with the number of ranks,
not just the query-side work.  Every rank searches its (query range, reference range) cell; the per-cell hit tables
(fixed-size records + CIGAR arena) are exchanged with ONE all-gather (RCCL over xGMI on the GPU box, gloo in the CPU tests),
after which every rank holds the whole table for the order-dependent host pass / the union-find.

Exactness.  The reference's ranking is per query and per database split (`-k 10` on each of 5 round-robin splits,
uberBlast.py:546-552): split = target index mod 5 over the WHOLE reference set.  A column therefore ranks with
split = (local target index + number of targets in the columns before it) mod 5 (`pep_search_params.t_index_base`), which
makes its local top-k a superset of its share of the global top-k; `pep_merge_hits` re-applies the top-k to the union and
orders the rows by (q, t) - the table is bit-identical to the single-GPU one (gloo tests with 2 and 4 ranks).
Pure query sharding (C = 1) needs no merge at all.

torch.distributed is plumbing only.  The reference has no counterpart (single-host multiprocessing with files as transport,
uberBlast.py:333-338, 500-503)."""
import numpy as np


def shard_bounds(lengths, world):
    """contiguous shards balanced by cumulative residue count; returns world+1 boundaries"""
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


def grid_shape(world):
    """(R query shards, C reference shards) with R * C == world.  The reference side carries most of the per-rank work that does not
    depend on the queries (6 frames per gene: K1 and the streaming of 6x more residues than the query side), so C >= R:
    1 -> 1x1, 2 -> 1x2, 4 -> 2x2, 8 -> 2x4, 16 -> 4x4; a world size without such a factorisation shards the reference only."""
    r = 1
    while (r * 2) * (r * 2) <= world and world % (r * 2) == 0:
        r *= 2
    return r, world // r


GRID_MIN_REF_NT = 20000000     # reference nucleotides from which the reference side is split as well (choose_grid)


def choose_grid(world, ref_nt_total):
    """The grid a sharded search uses by default.  Splitting the reference shrinks the part of a search that grows with it (the streaming
    of six frames per gene through the seed index) but costs the top-k merge after the exchange (0.3 ms for 35 k hits); below about
    20 Mnt of reference - measured per-rank cells of the 10 k-gene workload: 2x4 1.37 ms, 8x1 1.39 ms, 1x8 1.25 ms, all of it fixed cost -
    the merge is the larger of the two, and the queries alone are sharded (no merge: the tables concatenate)."""
    if ref_nt_total < GRID_MIN_REF_NT:
        return world, 1
    return grid_shape(world)


class _Staging(object):
    """byte buffers of one exchange, kept between steps and grown geometrically: page-locked host memory + device memory when the
    group runs over RCCL, plain host tensors under gloo.  A step then costs no allocation and no page faults."""

    def __init__(self, device):
        import torch
        self.device = device if device is not None else torch.device('cpu')
        self.on_gpu = self.device.type == 'cuda'
        self.bufs = {}

    def get(self, name, nbytes, host):
        import torch
        t = self.bufs.get(name)
        if t is None or t.numel() < nbytes:
            cap = max(1 << 16, int(nbytes * 1.5))
            if host:
                t = torch.empty(cap, dtype=torch.uint8, pin_memory=self.on_gpu)
            else:
                t = torch.empty(cap, dtype=torch.uint8, device=self.device)
            self.bufs[name] = t
        return t[:nbytes]


_staging = {}


def _staging_for(device):
    key = str(device)
    if key not in _staging:
        _staging[key] = _Staging(device)
    return _staging[key]


def _exchange(st, send, slot, world, group):
    """all-gather of one `slot`-byte block per rank through the staging buffers -> uint8 numpy view of world * slot bytes.
    send: this rank's block in the page-locked send buffer, or None when it has been written into the DEVICE send buffer already"""
    import torch
    import torch.distributed as dist
    recv = st.get('recv', slot * world, host=True)
    if st.on_gpu:
        d_send, d_recv = st.get('d_send', slot, host=False), st.get('d_recv', slot * world, host=False)
        if send is not None:
            d_send.copy_(send, non_blocking=True)
        dist.all_gather_into_tensor(d_recv, d_send, group=group)
        recv.copy_(d_recv, non_blocking=True)
        torch.cuda.current_stream(st.device).synchronize()
    else:
        dist.all_gather_into_tensor(recv, send, group=group)
    return recv.numpy()


def _all_gather_bytes(fill, nbytes, group, device, stage=None, hint=None, fill_dev=None):
    """all-gather of one variable-length byte string per rank: `fill(view)` writes this rank's `nbytes` bytes into a uint8 numpy view of
    the send buffer.  Returns (flat uint8 numpy view of the receive buffer, slot size, offset of the payload inside a slot, per-rank byte
    counts); rank r's bytes are flat[r * slot + offset : r * slot + offset + counts[r]].
    Two collectives in general - the byte counts, then the payloads padded to the largest.  With `hint` (a dict the caller keeps between
    calls) ONE: every rank sends a slot of the size that was enough last time (+ 25 %), its byte count in front; only when some rank's
    payload does not fit - every rank sees that in the gathered counts - the exchange is repeated the two-step way.
    fill_dev (RCCL groups only): `fill_dev(tensor)` writes the payload into a uint8 DEVICE tensor instead - the payload is on the GPU
    already (a hit table left there by the search) and goes from device memory straight into the collective, without the round trip
    through the host that `fill` implies."""
    import torch
    import torch.distributed as dist
    st = stage if stage is not None else _staging_for(device)
    world = dist.get_world_size(group)
    on_dev = fill_dev is not None and st.on_gpu
    if hint is not None and hint.get('slot', 0) >= 16:
        slot = int(hint['slot'])
        if on_dev:
            head = st.get('head', 8, host=True)
            head.numpy().view(np.uint64)[0] = nbytes
            d_send = st.get('d_send', slot, host=False)
            d_send[:8].copy_(head, non_blocking=True)
            if 8 + nbytes <= slot:
                fill_dev(d_send[8:8 + nbytes])
            send = None
        else:
            send = st.get('send', slot, host=True)
            view = send.numpy()
            view[:8].view(np.uint64)[0] = nbytes
            if 8 + nbytes <= slot:
                fill(view[8:8 + nbytes])
        flat = _exchange(st, send, slot, world, group)
        counts = np.array([int(flat[r * slot:r * slot + 8].view(np.uint64)[0]) for r in range(world)], dtype=np.int64)
        if int(counts.max()) + 8 <= slot:
            return flat, slot, 8, counts
    n = torch.tensor([nbytes], dtype=torch.int64, device=st.device)
    counts = torch.empty(world, dtype=torch.int64, device=st.device)
    dist.all_gather_into_tensor(counts, n, group=group)
    counts = counts.cpu().numpy()
    slot = (int(counts.max()) + 255) & ~255
    if hint is not None:
        hint['slot'] = ((int(counts.max()) * 5 // 4 + 8 + 255) & ~255)
    if slot == 0:
        return np.zeros(0, dtype=np.uint8), 0, 0, counts
    if on_dev:
        fill_dev(st.get('d_send', slot, host=False)[:nbytes])
        return _exchange(st, None, slot, world, group), slot, 0, counts
    send = st.get('send', slot, host=True)
    fill(send.numpy()[:nbytes])
    return _exchange(st, send, slot, world, group), slot, 0, counts


class _DeviceBytes(object):
    """raw device memory as something torch.as_tensor understands (the library hands out plain addresses, include/peppan_hip.h)"""

    def __init__(self, address, nbytes):
        self.__cuda_array_interface__ = dict(shape=(int(nbytes),), typestr='|u1', data=(int(address), False), version=2)


def allgather_hits(hits, cigar, q_base, t_base=0, group=None, device=None, out=None, hint=None, on_device=None, force=False):
    """hits: structured array (peppan_amd._native.HIT_DTYPE) with shard-local q / t indices; cigar: uint32 arena.
    Returns the concatenated (hits, cigar) of all ranks in rank order with global indices and re-based cigar offsets.
    Collectives: one all-gather of the payload sizes, one of a padded payload (8-byte hit count + hit records + arena).
    `out`: optional dict that keeps the output arrays between calls (the result is then only valid until the next call).
    `hint`: optional dict kept between calls: repeated exchanges of similar size then need ONE collective (_all_gather_bytes).
    `on_device`: (n_hits, n_cigar, address of the hit records, address of the arena) of a table the search left ON THE GPU
    (Context.search_on_device) - hits / cigar are then not looked at: the payload is assembled in device memory (record copy, q / t
    re-based by two strided adds) and handed to the collective from there.
    `force`: run the collectives even in a one-rank group (tests and tools on a one-GPU box)."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        if on_device is not None:
            raise ValueError('allgather_hits: on_device needs an initialised process group')
        if q_base or t_base:
            hits = hits.copy()
            hits['q'] += q_base
            hits['t'] += t_base
        return hits, cigar
    from . import _native as N
    rec = N.HIT_DTYPE.itemsize
    nh, nc = (len(hits), len(cigar)) if on_device is None else (int(on_device[0]), int(on_device[1]))

    def fill(view):
        view[:8].view(np.uint64)[0] = nh
        mine = view[8:8 + nh * rec].view(N.HIT_DTYPE)
        mine[:] = hits
        mine['q'] += q_base
        mine['t'] += t_base
        view[8 + nh * rec:].view(np.uint32)[:] = cigar

    fill_dev = None
    if on_device is not None:
        import torch
        st = _staging_for(device)
        if not st.on_gpu:
            raise ValueError('allgather_hits: on_device needs an RCCL group (device tensors)')

        def fill_dev(t):
            head = st.get('nh', 8, host=True)
            head.numpy().view(np.uint64)[0] = nh
            t[:8].copy_(head, non_blocking=True)
            if nh:
                t[8:8 + nh * rec].copy_(torch.as_tensor(_DeviceBytes(on_device[2], nh * rec), device=st.device))
                cols = t[8:8 + nh * rec].view(torch.int32).view(nh, rec // 4)          # q and t are the first two 32-bit fields of a record
                if q_base:
                    cols[:, 0] += int(q_base)
                if t_base:
                    cols[:, 1] += int(t_base)
            if nc:
                t[8 + nh * rec:8 + nh * rec + 4 * nc].copy_(torch.as_tensor(_DeviceBytes(on_device[3], 4 * nc), device=st.device))

    flat, slot, at, counts = _all_gather_bytes(fill, 8 + nh * rec + 4 * nc, group, device, hint=hint, fill_dev=fill_dev)
    parts, tot_h, tot_c = [], 0, 0
    for r in range(len(counts)):
        part = flat[r * slot + at:r * slot + at + int(counts[r])]
        n_r = int(part[:8].view(np.uint64)[0])
        h, c = part[8:8 + n_r * rec].view(N.HIT_DTYPE), part[8 + n_r * rec:].view(np.uint32)
        parts.append((h, c))
        tot_h, tot_c = tot_h + n_r, tot_c + len(c)
    if out is None:
        all_h, all_c = np.empty(tot_h, dtype=N.HIT_DTYPE), np.empty(tot_c, dtype=np.uint32)
    else:
        if len(out.get('h', ())) < tot_h:
            out['h'] = np.empty(int(tot_h * 1.5) + 64, dtype=N.HIT_DTYPE)
        if len(out.get('c', ())) < tot_c:
            out['c'] = np.empty(int(tot_c * 1.5) + 64, dtype=np.uint32)
        all_h, all_c = out['h'][:tot_h], out['c'][:tot_c]
    ph = pc = 0
    for h, c in parts:
        dst = all_h[ph:ph + len(h)]
        dst[:] = h
        dst['cigar_off'] += pc
        all_c[pc:pc + len(c)] = c
        ph, pc = ph + len(h), pc + len(c)
    return all_h, all_c


class ShardedSearch(object):
    """The all-vs-all translated search of `query_nt` against `ref_nt` (lists of nucleotide byte strings in the reference's FASTA
    order) over the ranks of `group` (None + world 1 = single process).  `ctx` is anything with the interface of
    peppan_amd._native.Context (the tests pass the oracle-backed context).

        s = ShardedSearch(ctx, query_nt, ref_nt, params, rank, world)
        hits, cigar, stats = s.search()        # the WHOLE table on every rank: global q / t indices, rows by (q, t)
        s.gene_of_target                       # reference sequence index of every global target (for the single-linkage step)
    """

    def __init__(self, ctx, query_nt, ref_nt, params, rank=0, world=1, frames=6, gtable=11, grid=None, group=None, device=None):
        self.ctx, self.params, self.rank, self.world, self.group, self.device = ctx, params, rank, world, group, device
        self.R, self.C = grid if grid is not None else choose_grid(world, sum(len(x) for x in ref_nt))
        if self.R * self.C != world:
            raise ValueError('grid %dx%d does not match world size %d' % (self.R, self.C, world))
        self.r, self.c = divmod(rank, self.C)
        qb = shard_bounds([len(s) for s in query_nt], self.R)
        gb = shard_bounds([len(s) for s in ref_nt], self.C)
        self.q0, self.q1, self.g0, self.g1 = qb[self.r], qb[self.r + 1], gb[self.c], gb[self.c + 1]
        self.n_queries, self.n_refs = len(query_nt), len(ref_nt)
        ctx.set_query_nt(query_nt[self.q0:self.q1], gtable)
        ctx.set_ref_nt(ref_nt[self.g0:self.g1], frames, gtable)
        ctx.translate()
        local_gene = ctx.target_meta()['seq'].astype(np.uint32) + np.uint32(self.g0)
        if world == 1:
            self.t_base, self.gene_of_target = 0, local_gene
        else:
            # what K1 made of every column's genes: the number of targets in front of this column and the global target -> gene map
            import torch
            dev = device if device is not None else torch.device('cpu')
            raw = local_gene.view(np.uint8)
            flat, slot, at, counts = _all_gather_bytes(lambda view: view.__setitem__(slice(None), raw), raw.size, group, dev)
            cols = [flat[c * slot + at:c * slot + at + int(counts[c])].view(np.uint32).copy() for c in range(self.C)]   # row 0 holds one rank per column, in column order
            self.t_base = int(sum(len(x) for x in cols[:self.c]))
            self.gene_of_target = np.concatenate(cols)
        self.params.t_index_base = self.t_base
        self._scratch, self._merged, self._hint = {}, {}, {}

    def search(self, retranslate=False, copy=True):
        import time
        t0 = time.perf_counter()
        if retranslate:
            self.ctx.invalidate_translation() if hasattr(self.ctx, 'invalidate_translation') else self.ctx.translate(force=True)
        t_k1 = time.perf_counter()
        keep = None if copy else self._scratch            # copy=False: the arrays of the previous step are overwritten
        if self.world > 1 and self.device is not None and getattr(self.device, 'type', '') == 'cuda' and hasattr(self.ctx, 'search_on_device'):
            # RCCL: the table never leaves the GPU before the collective (no copy to the host at the end of the search, none back up)
            nh, nc, stats, ptrs = self.ctx.search_on_device(self.params)
            t1 = time.perf_counter()
            allh, allc = allgather_hits(None, None, self.q0, self.t_base, group=self.group, device=self.device, out=keep, hint=self._hint,
                                        on_device=(nh, nc) + tuple(ptrs))
        else:
            hits, cigar, stats = self.ctx.search(self.params, copy=(copy and self.world == 1))
            t1 = time.perf_counter()
            allh, allc = allgather_hits(hits, cigar, self.q0, self.t_base, group=self.group, device=self.device, out=keep, hint=self._hint)
        t2 = time.perf_counter()
        if self.C > 1:
            from . import _native as N
            allh, allc = N.merge_hits(allh, allc, self.params.top_k, self.params.n_splits, out=None if copy else self._merged)
        # host wall time of the three parts of a step (the device phases are in the pep_search_stats fields)
        stats = dict(stats, ms_host_translate=(t_k1 - t0) * 1e3, ms_host_search=(t1 - t0) * 1e3, ms_host_exchange=(t2 - t1) * 1e3, ms_host_merge=(time.perf_counter() - t2) * 1e3)
        return allh, allc, stats
