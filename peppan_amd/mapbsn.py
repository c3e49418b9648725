"""The caller and the on-disk formats on the far side of the genes->genomes mapping search.

    MapBsn              PEPPAN.py:27-114    zip archive whose members are .npy payloads keyed by str(key)
    decodeSeq           PEPPAN.py:318-324   base-5 triple packing of the aligned-allele strings
    compare_prediction  PEPPAN.py:869-901   overlap of every hit with the genome's original annotation (column 10)
    iter_map_bsn        PEPPAN.py:759-867   one genome: search + grouping of merged fragments + allele strings + overlap classes
    get_map_bsn         PEPPAN.py:907-989   all genomes -> the four stores (.tab / .seq / .mat / .conflicts)

The reference runs one forked worker and one uberBlast call per genome (PEPPAN.py:922).  Here `get_map_bsn` hands MANY
genomes to one GPU search (`uberBlastBatch`, pep_set_target_groups keeps the per-genome ranking) and then performs the
same per-genome bookkeeping, in genome order, so the stores are those of the reference's in-order `map` variant
(PEPPAN.py:923).  Everything below the search is host logic pinned by tests/golden/g14_mapbsn.json / g15_getmapbsn.json.
"""
import io
import queue
import threading
import os
import re
import sys
import zipfile

import numpy as np

from .configure import logger

__all__ = ['MapBsn', 'decodeSeq', 'encodeSeq', 'compare_prediction', 'build_bsn', 'iter_map_bsn', 'get_map_bsn']


def _npy_bytes(val):
    buf = io.BytesIO()
    np.lib.format.write_array(buf, np.asanyarray(val), allow_pickle=True)
    return buf.getvalue()


class MapBsn(object):
    """dict-like store: zip member `str(key)` holds one array in .npy format (object arrays pickled).  Readable by
    `np.load(fname, allow_pickle=True)` like the reference's files (PEPPAN.py:1931)."""

    def __init__(self, fname, mode='r'):
        self.fname, self.mode = fname, mode
        self.conn = zipfile.ZipFile(fname, mode=mode, compression=zipfile.ZIP_DEFLATED, allowZip64=True, compresslevel=1)
        self.namelist = set(self.conn.namelist())
        # writes go through ONE background thread per store, in order: deflate releases the GIL, so the compression of a genome's
        # tables overlaps with the Python bookkeeping of the next one (a third of get_map_bsn's time was spent inside zlib)
        self._queue = self._thread = self._error = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        try:
            self._flush()
        finally:
            if self._thread is not None:
                self._queue.put(None)
                self._thread.join()
                self._thread = None
            self.conn.close()

    def _writer(self):
        while True:
            item = self._queue.get()
            try:
                if item is None:
                    return
                db, key, data = item
                if self._error is None:
                    # members are read back whole either way; deflating a few hundred bytes costs more than it saves (zlib set-up per member)
                    db.writestr(key, data, compress_type=zipfile.ZIP_STORED if len(data) < 4096 else None)
            except BaseException as e:              # reported by the next _flush() on the owning thread
                self._error = e
            finally:
                self._queue.task_done()

    def _flush(self):
        """wait until every queued member is in its archive"""
        if self._thread is not None:
            self._queue.join()
        if self._error is not None:
            e, self._error = self._error, None
            raise e

    def exists(self, key):
        return str(key) in self.namelist

    def get(self, key, default=[]):
        key = str(key)
        if key not in self.namelist:
            return default
        self._flush()
        return np.lib.format.read_array(io.BytesIO(self.conn.read(key)), allow_pickle=True)

    __getitem__ = get

    def keys(self):
        return self.namelist

    def values(self):
        for key in self.namelist:
            yield self.get(key)

    def items(self):
        for key in self.namelist:
            yield key, self.get(key)

    def size(self):
        return len(self.namelist)

    def delete(self, key):
        """logical delete: the member stays in the archive but is no longer listed"""
        self.namelist.discard(str(key))

    def pop(self, key, default=[]):
        val = self.get(key, default)
        self.delete(key)
        return val

    def delete_real(self, key):
        """physically drop a member.  The reference shells out to `zip -d` and reopens the archive with mode 'w'
        (PEPPAN.py:71-77), which truncates it; here the archive is rewritten in-process without the member and every other
        member is kept."""
        key = str(key)
        if key not in self.namelist:
            return
        self._flush()
        self.namelist.discard(key)
        self.conn.close()
        tmp = self.fname + '.rewrite'
        with zipfile.ZipFile(self.fname) as src, zipfile.ZipFile(tmp, 'w', compression=zipfile.ZIP_DEFLATED, allowZip64=True, compresslevel=1) as dst:
            for name in src.namelist():
                if name != key:
                    dst.writestr(name, src.read(name))
        os.replace(tmp, self.fname)
        self.conn = zipfile.ZipFile(self.fname, mode='a', compression=zipfile.ZIP_DEFLATED, allowZip64=True, compresslevel=1)

    def _save(self, db, key, val):
        data = _npy_bytes(val)                      # serialised here: the caller may change `val` afterwards
        if self._thread is None:
            self._queue = queue.Queue(maxsize=256)
            self._thread = threading.Thread(target=self._writer, daemon=True)
            self._thread.start()
        self._queue.put((db, key, data))

    def save(self, key, val):
        key = str(key)
        self.delete_real(key)
        self._save(self.conn, key, val)
        self.namelist.add(key)

    def update(self, dataset):
        """merge a list of 2-D arrays, each keyed by its [0][0], into the store (rows appended to what the key holds)"""
        tmp_name = self.fname[:-4] + '.tmp.npz'
        seen = set()
        with zipfile.ZipFile(tmp_name, mode='w', compression=zipfile.ZIP_DEFLATED, allowZip64=True, compresslevel=1) as tmp:
            for d in dataset:
                key = str(d[0][0])
                seen.add(key)
                old = self.get(key)
                self._save(tmp, key, np.vstack([old, d]) if len(old) else d)
            for key in list(self.keys()):
                if key not in seen:
                    data = self.get(key)
                    if len(data):
                        seen.add(key)
                        self._save(tmp, key, data)
            self._flush()
        self.conn.close()
        self.namelist = seen
        os.rename(tmp_name, self.fname)
        self.conn = zipfile.ZipFile(self.fname, mode='a', compression=zipfile.ZIP_DEFLATED, allowZip64=True, compresslevel=1)


# ------------------------------------------------------------------------------------------------ allele strings
_BASE = np.zeros(256, dtype=np.uint8)
_BASE[[ord(c) for c in 'ACGT']] = (1, 2, 3, 4)


def encodeSeq(b):
    """uint8 base codes (0 gap/unknown, 1..4 ACGT) of length L -> ceil(L/3) bytes: first third * 25 + second third * 5 +
    last third (zero padded), the packing at PEPPAN.py:847-848"""
    s = -(-b.shape[0] // 3)
    tail = np.concatenate([b, np.zeros(-b.shape[0] % 3, dtype=int)])[2 * s:]
    return (b[:s] * 25 + b[s:2 * s] * 5 + tail).astype(np.uint8)


def decodeSeq(seqs):
    """inverse of encodeSeq for a [n, s] matrix (PEPPAN.py:318-324): [n, 3s] codes"""
    n, s = seqs.shape
    out = np.zeros([n, s * 3], dtype=np.uint8)
    out[:, :s] = seqs // 25
    out[:, s:2 * s] = (seqs % 25) // 5
    out[:, 2 * s:] = seqs % 5
    return out


# ------------------------------------------------------------------------------------------------ old annotation
def _stable_order(tab, keys):
    """row order of a stable multi-key sort (first key most significant), like DataFrame.sort_values(by=keys)"""
    cols = []
    for k in reversed(keys):
        c = (k if isinstance(k, np.ndarray) else tab[:, k]).tolist()
        if any(isinstance(v, str) for v in c):
            cols.append(np.unique(np.array(c, dtype=str), return_inverse=True)[1])
        elif all(isinstance(v, (int, np.integer)) for v in c):
            cols.append(np.asarray(c, dtype=np.int64))
        else:
            cols.append(np.asarray(c, dtype=np.float64))
    return np.lexsort(cols)


def compare_prediction(blastab, old_prediction):
    """column 10 <- the largest fraction of an in-frame, same-strand original gene that a hit covers (0.1 if none);
    returns the table sorted by (query, contig, score).  PEPPAN.py:869-901."""
    lo = np.minimum(blastab[:, 8].astype(np.int64), blastab[:, 9].astype(np.int64))
    blastab = blastab[_stable_order(blastab, [1, lo])]
    blastab[:, 10] = 0.1
    with MapBsn(old_prediction) as op:
        contig, genes, at = None, [], 0
        for bsn in blastab:
            if contig != bsn[1]:
                contig, genes, at = bsn[1], op.get(bsn[1]), 0
            head, tail = bsn[8] - bsn[6] + 1, bsn[9] + (bsn[12] - bsn[7])
            if bsn[8] < bsn[9]:
                s, e = bsn[8], bsn[9]
                frames = {head % 3 + 1, (tail + 1) % 3 + 1}
            else:
                s, e = bsn[9], bsn[8]
                frames = {(-head) % 3 - 1, (-(tail - 1)) % 3 - 1}
            while at < len(genes) and s > genes[at][2]:
                at += 1
            for p in genes[at:]:
                if e < p[1]:
                    break
                if p[3] == '+':
                    if p[1] % 3 + 1 not in frames and (p[2] + 1) % 3 + 1 not in frames:
                        continue
                elif (-(p[1] - 1)) % 3 - 1 not in frames and (-p[2]) % 3 - 1 not in frames:
                    continue
                plen = p[2] - p[1] + 1
                ovl = min(e, p[2]) - max(s, p[1]) + 1.
                if ovl >= 0.6 * plen or ovl >= 0.6 * (e - s + 1):
                    ovl = ovl / plen
                    if ovl > bsn[10]:
                        bsn[10] = ovl
    return blastab[_stable_order(blastab, [0, 1, 11])]


# ------------------------------------------------------------------------------------------------ one genome
def _passes(length, ql, params):
    return (length >= max(params['match_prop'] * ql, params['match_len']) or
            length >= max(params['match_prop1'] * ql, params['match_len1']) or
            length >= max(params['match_prop2'] * ql, params['match_len2']))


def _encode_cigar_strings(cigars):
    """['60M3D90M', ...] -> (uint32 runs len<<2|op with op 0=M 1=I 2=D, runs per string) with two regex passes over one joined string"""
    joined = '|'.join(cigars) + '|'
    lens = np.array(re.findall(r'\d+', joined), dtype=np.int64)
    ops = np.frombuffer(re.sub(r'\d+', '', joined).encode('ascii'), dtype=np.uint8)
    sep = ops == ord('|')
    code = np.full(256, 3, dtype=np.uint32)
    code[[ord('M'), ord('I'), ord('D')]] = (0, 1, 2)
    runs = (lens.astype(np.uint32) << 2) | code[ops[~sep]]
    per = np.diff(np.concatenate([[-1], np.nonzero(sep)[0]])) - 1
    return runs, per


def build_bsn(blastab, overlap, seq, orthoGroup, old_prediction, params, ctx=None):
    """(17-column table with merge groups, int[m,3] overlaps) of ONE genome -> (bsn object[n,7], ovl int[k,3]).
    bsn row = [gene, contig, score, identity, packed allele, group id, rows(object[k,16])].  PEPPAN.py:773-866.
    The per-hit allele strings, their in-frame / stop-free lengths and the packing run on the GPU (K12, `ctx.alleles`)."""
    if blastab.shape[0] == 0:
        return np.empty([0, 7], dtype=object), np.zeros([0, 3], dtype=np.int64)
    if ctx is None:
        from .uberBlast import get_context
        ctx = get_context()
    blastab.T[:2] = blastab.T[:2].astype(int)
    blastab = compare_prediction(blastab, old_prediction)
    n_id = int(np.max(blastab.T[15])) + 1
    kept = np.zeros(n_id, dtype=bool)
    single, chained = [], {}
    mi = params['match_identity']
    for tab in blastab:
        grp = tab[16]
        if not (grp[1] >= mi and _passes(grp[2], tab[12], params)):
            tab[2] = -1
            continue
        kept[tab[15]] = True
        if len(grp) <= 4:
            single.append([tab[0], tab[1], grp[0], grp[1], None, 0, [tab]])
            continue
        if tab[2] >= mi and _passes(tab[7] - tab[6] + 1, tab[12], params):
            single.append([tab[0], tab[1], tab[11], tab[2], None, 0, [tab]])
        members = grp[3:]
        if grp[3] not in chained:
            chained[grp[3]] = [tab[0], tab[1], grp[0], grp[1], None, 0, [[]] * len(members)]
        chained[grp[3]][6][members.index(tab[15])] = tab
    groups = single + list(chained.values())
    overlap = overlap[kept[overlap.T[0]] & kept[overlap.T[1]], :2]
    # ---- K12 over every row of every group
    flat = [tab for group in groups for tab in group[6]]
    n_rows = np.array([len(group[6]) for group in groups], dtype=np.int64)
    grp_off = np.concatenate([[0], np.cumsum(n_rows)]).astype(np.uint64)
    contigs = [s for n, s in seq]
    cidx = {n: i for i, (n, s) in enumerate(seq)}
    runs, per = _encode_cigar_strings([tab[14] for tab in flat])
    from ._native import LOCUS_DTYPE
    loci = np.zeros(len(flat), dtype=LOCUS_DTYPE)
    loci['contig'] = [cidx[tab[1]] for tab in flat]
    cols = np.array([[tab[6], tab[7], tab[8], tab[9], tab[12], tab[15]] for tab in flat], dtype=np.int64).reshape(-1, 6)
    loci['q_start'], loci['rs'], loci['re'] = cols[:, 0], cols[:, 2], cols[:, 3]
    loci['cigar_runs'] = per
    loci['cigar_off'] = np.concatenate([[0], np.cumsum(per)[:-1]]) if len(per) else []
    loci['group'] = np.repeat(np.arange(len(groups)), n_rows)
    first = grp_off[:-1].astype(np.int64)
    in_frame, orf, packed = ctx.alleles(contigs, loci, runs, grp_off, cols[first, 4], params['gtable'])
    sc = np.minimum(in_frame, orf + 3)
    iden = np.array([tab[2] for tab in flat], dtype=np.float64)
    known = np.array([tab[10] for tab in flat], dtype=np.float64)
    qspan = cols[:, 1] - cols[:, 0] + 1
    r = np.sqrt(sc.astype(np.float64) / cols[:, 4] * known)
    msc = (sc * iden) * np.sqrt(sc * r)
    amsc = msc / qspan
    pack_off = np.concatenate([[0], np.cumsum((cols[first, 4] + 2) // 3)])
    # ---- assemble
    as_single, as_chain = np.full(n_id, -1, dtype=np.int64), np.full(n_id, -1, dtype=np.int64)
    gid_of_row = loci['group'].astype(np.int64)
    multi = np.repeat(n_rows > 1, n_rows)
    as_single[cols[~multi, 5]] = gid_of_row[~multi]
    as_chain[cols[multi, 5]] = gid_of_row[multi]
    bsn = np.empty([len(groups), 7], dtype=object)
    for gid, group in enumerate(groups):
        lo, hi = int(grp_off[gid]), int(grp_off[gid + 1])
        if hi - lo == 1:
            score = msc[lo]
            rows = group[6][0][:16].reshape(1, 16)
        else:
            spans = [[cols[k, 0], cols[k, 1], amsc[k], msc[k]] for k in range(lo, hi)]
            for prev, cur in zip(spans[:-1], spans[1:]):          # fragments overlapping on the query: the weaker one is trimmed
                if cur[0] < prev[1]:
                    if cur[2] > prev[2]:
                        prev[1] = cur[0] - 1
                        prev[3] = prev[2] * (prev[1] - prev[0] + 1)
                    else:
                        cur[0] = prev[1] + 1
                        cur[3] = cur[2] * (cur[1] - cur[0] + 1)
            score = np.sum([c[3] for c in spans])
            rows = np.array([tab[:16] for tab in group[6]])
        row = bsn[gid]
        row[0], row[1], row[2], row[3], row[4], row[5], row[6] = group[0], group[1], score, group[3], packed[pack_off[gid]:pack_off[gid + 1]], gid, rows
    a0, c0, a1, c1 = as_single[overlap.T[0]], as_chain[overlap.T[0]], as_single[overlap.T[1]], as_chain[overlap.T[1]]
    overlap = np.vstack([np.vstack([m, n]).T[(m >= 0) & (n >= 0)] for m in (a0, c0) for n in (a1, c1)] +
                        [np.vstack([as_single, as_chain]).T[(as_single >= 0) & (as_chain >= 0)]])
    if overlap.shape[0]:
        og = np.load(orthoGroup, allow_pickle=True) if isinstance(orthoGroup, str) else orthoGroup
        rel = {}
        for g in og[og.T[2] != 0]:
            rel[(g[0], g[1])] = 1 if g[2] > 0 else -1
        for g in og[og.T[2] != 0]:
            rel[(g[1], g[0])] = 1 if g[2] > 0 else -1
        score = np.array([0 if m == n else rel.get((m, n), 2) for m, n in zip(bsn[overlap.T[0], 0], bsn[overlap.T[1], 0])], dtype=np.int64)
        overlap = np.hstack([overlap, score[:, np.newaxis]])[score >= 0]
    else:
        overlap = np.zeros([0, 3], dtype=np.int64)
    return bsn, overlap


def _map_argv(clust, params):
    tools = '--blastn' if params.get('noDiamond') else '--blastn --diamond'
    tail = '-t 1 -e 0,3' if params.get('noDiamond') else '-t 1 -s 1 -e 0,3'
    return '-q {0} -f -m -O {1} --min_id {2} --min_cov {3} --min_ratio {4} --merge_gap {5} --merge_diff {6} {7} --gtable {8}'.format(
        clust, tools, params['match_identity'] - 0.1, params['match_frag_len'], params['match_frag_prop'], params['link_gap'],
        params['link_diff'], tail, params['gtable']).split()


def _write_genome(prefix, id, seq):
    gfile = '{0}.{1}.genome'.format(prefix, id)
    with open(gfile, 'w') as fout:
        for n, s in seq:
            fout.write('>{0}\n{1}\n'.format(n, s))
    return gfile


def iter_map_bsn(data):
    """one genome, the reference's worker signature (PEPPAN.py:759-867): writes `<prefix>.<id>.bsn.npz`, returns its prefix"""
    from .uberBlast import uberBlast
    prefix, clust, id, taxon, seq, orthoGroup, old_prediction, params = data
    gfile = _write_genome(prefix, id, seq)
    try:
        blastab, overlap = uberBlast(['-r', gfile] + _map_argv(clust, params))
    finally:
        os.unlink(gfile)
    bsn, ovl = build_bsn(blastab, overlap, seq, orthoGroup, old_prediction, params)
    out_prefix = '{0}.{1}'.format(prefix, id)
    np.savez_compressed(out_prefix + '.bsn.npz', bsn=bsn, ovl=ovl)
    return out_prefix


# ------------------------------------------------------------------------------------------------ all genomes
CHUNK = 1000          # arrays per member of the .seq / .mat stores (PEPPAN.py:953, 962)
BLOCK = 30000         # group ids per member of the .conflicts store (PEPPAN.py:934-947)


def _gpu_search(prefix, clust, jobs, params, genomes_per_batch=64):
    """yield (blastab, overlap) per genome, `genomes_per_batch` genomes per GPU search"""
    from .uberBlast import uberBlastBatch
    argv = _map_argv(clust, params)
    for lo in range(0, len(jobs), genomes_per_batch):
        files = [_write_genome(prefix, id, seq) for id, taxon, seq in jobs[lo:lo + genomes_per_batch]]
        try:
            results = uberBlastBatch(files, argv)
        finally:
            for f in files:
                os.unlink(f)
        for r in results:
            yield r


def _all_bsn(prefix, clust, jobs, og, old_prediction, params, search, ctx, group, per_round):
    """(job, (bsn, ovl)) for every genome in job order.  With torch.distributed initialised the genomes are dealt to the ranks in
    blocks of `per_round` (independent units, no data-path collective); every rank maps its block on its own GPU and rank 0
    gathers the finished per-genome objects - only rank 0 yields, the others just take part."""
    def local(mine):
        out = []
        for (id, taxon, seq), (blastab, overlap) in zip(mine, search(prefix, clust, mine, params) if mine else ()):
            out.append(build_bsn(blastab, overlap, seq, og, old_prediction, params, ctx))
        return out
    world, rank = 1, 0
    dist = sys.modules.get('torch.distributed')            # only a caller that set up a process group has imported it
    if dist is not None and dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    if world == 1:
        for job, (blastab, overlap) in zip(jobs, search(prefix, clust, jobs, params)):
            yield job, build_bsn(blastab, overlap, job[2], og, old_prediction, params, ctx)
        return
    n_rounds = -(-len(jobs) // (per_round * world))
    for k in range(n_rounds):
        lo = (k * world + rank) * per_round
        done = local(jobs[lo:lo + per_round])
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(done, gathered, dst=0, group=group)
        if rank == 0:
            for r in range(world):
                lo_r = (k * world + r) * per_round
                for job, out in zip(jobs[lo_r:lo_r + per_round], gathered[r]):
                    yield job, out


def get_map_bsn(prefix, clust, genomes, orthoGroup, old_prediction, conn, seq_conn, mat_conn, clf_conn, saveSeq, params, search=None, ctx=None,
                group=None, genomes_per_round=64):
    """genomes: {contig id: [taxon id, sequence]} -> fills the four MapBsn stores like PEPPAN.py:907-989:
      conn      gene id -> int rows [gene, taxon, score*1e4, ident*1e4, ident*1e4, group id, n fragments], best score first
      seq_conn  chunk no -> object array of packed alleles (only with saveSeq)
      mat_conn  chunk no -> object array of the hit rows of each group
      clf_conn  block no -> CSR [30001 offsets + 30001, partner*10 + class] of group-overlap conflicts
    `search(prefix, clust, jobs, params)` yields (blastab, overlap) per genome in job order (default: batched GPU search).
    Under torch.distributed (one process per GPU) the genomes are sharded over the ranks in blocks of `genomes_per_round`; rank 0
    writes the stores (the other ranks pass None for them), whose contents do not depend on the number of ranks."""
    if len(genomes) == 0:
        raise ValueError('get_map_bsn: no genome to map against')
    taxa = {}
    for g, s in genomes.items():
        taxa.setdefault(s[0], []).append([g, s[1]])
    jobs = [(id, taxon, seq) for id, (taxon, seq) in enumerate(taxa.items())]
    og = np.load(orthoGroup, allow_pickle=True)
    n_group = 0
    seqs, seq_cnt = [], 0
    mats, mat_cnt = [], 0
    tabs, conflicts = [], {}

    def flush_conflicts(block):
        ovl = np.vstack(conflicts.pop(block))
        clf_conn.save(block, np.concatenate([np.cumsum(np.concatenate([[0], np.bincount(ovl.T[0], minlength=BLOCK)])) + BLOCK + 1, ovl.T[1]]))

    per_round = max(1, int(genomes_per_round))
    searcher = search or (lambda *a: _gpu_search(*a, genomes_per_batch=per_round))
    for (id, taxon, seq), (bsn, ovl) in _all_bsn(prefix, clust, jobs, og, old_prediction, params, searcher, ctx, group, per_round):
        bId = id
        last = bId == len(jobs) - 1
        if bsn.shape[0]:
            bsn.T[5] += n_group
            ovl[:, :2] += n_group
            first, n_group = n_group, n_group + bsn.shape[0]
            bsn.T[1] = genomes.get(bsn[0, 1], [-1])[0]
            if ovl.shape[0]:
                for block in np.unique((ovl[:, :2] / BLOCK).astype(int)):
                    conflicts.setdefault(block, [])
                ovl = np.vstack([ovl, ovl[:, (1, 0, 2)]])
                ovl = ovl[np.argsort(ovl.T[0])]
                ovl = np.hstack([(ovl[:, :1] / BLOCK).astype(int), ovl[:, :1] % BLOCK, ovl[:, 1:2] * 10 + ovl[:, 2:]])
                for part in np.split(ovl, np.cumsum(np.unique(ovl.T[0], return_counts=True)[1])[:-1]):
                    conflicts[part[0, 0]].append(part[:, 1:])
                for block in np.arange(int(first / BLOCK), int(n_group / BLOCK)):
                    if block in conflicts:
                        flush_conflicts(block)
            if saveSeq:
                seqs = np.concatenate([seqs, bsn.T[4]])
                pieces = np.split(seqs, np.arange(CHUNK, seqs.shape[0], CHUNK))
                seqs = pieces[-1]
                for s in pieces[:-1]:
                    seq_conn.save(seq_cnt, s)
                    seq_cnt += 1
            bsn.T[4] = bsn.T[3]
            mats = np.concatenate([mats, bsn.T[6]])
            pieces = np.split(mats, np.arange(CHUNK, mats.shape[0], CHUNK))
            mats = pieces[-1]
            for m in pieces[:-1]:
                mat_conn.save(mat_cnt, m)
                mat_cnt += 1
            bsn.T[6] = np.array([len(b) for b in bsn.T[6]], dtype=np.uint8)
            bsn.T[2:5] = bsn.T[2:5] * 10000
            tabs.append(bsn[np.argsort(-bsn.T[2])].astype(int))
        logger('Merged {0}.{1}'.format(prefix, id))
        if (bId % 500 == 499 or last) and len(tabs):
            tab = np.vstack(tabs)
            tab = tab[np.argsort(tab.T[0], kind='mergesort')]
            conn.update(np.split(tab, np.cumsum(np.unique(tab.T[0], return_counts=True)[1])[:-1]))
            tabs = []
    if saveSeq and len(seqs):
        seq_conn.save(seq_cnt, seqs)
    if len(mats):
        mat_conn.save(mat_cnt, mats)
    for block in list(conflicts.keys()):
        flush_conflicts(block)
