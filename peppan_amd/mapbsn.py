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
import contextlib
import io
import queue
import threading
import os
import sys
import time
import zipfile
import zlib

import numpy as np

from .configure import logger, effective_cpus

__all__ = ['MapBsn', 'decodeSeq', 'encodeSeq', 'compare_prediction', 'GenomeGroups', 'OrthoRelation', 'build_groups', 'build_bsn', 'iter_map_bsn', 'get_map_bsn']


def _npy_bytes(val):
    buf = io.BytesIO()
    np.lib.format.write_array(buf, np.asanyarray(val), allow_pickle=True)
    return buf.getvalue()


# ---- archive members, deflated off the writer's thread -------------------------------------------------------------------------------------
# zipfile deflates inside ZipFile.writestr, under the archive's lock: one member at a time per archive, 13 ms for the 0.5 MB of hit rows a
# genome adds to the .mat store - the store's writer thread then is what the whole mapping waits for.  Members are therefore made and
# deflated by a small pool shared by all stores (zlib releases the GIL) and the writer thread only appends finished payloads, in order:
# local header + payload at the archive's end, and the entry for the central directory that ZipFile.close() writes.
_PACKERS = None


def _packers():
    global _PACKERS
    if _PACKERS is None:
        from concurrent.futures import ThreadPoolExecutor
        _PACKERS = ThreadPoolExecutor(max_workers=int(os.environ.get('PEPPAN_PACK_THREADS', 0)) or max(2, min(16, effective_cpus() // 2)), thread_name_prefix='mapbsn-pack')
    return _PACKERS


FAST_DEFLATE = -1          # a "strategy" of _pack_member beside zlib's: the library's own coder with matches (csrc/stores.hip: pep_deflate_fast)


def _pack_member(data, strategy=zlib.Z_DEFAULT_STRATEGY):
    """data (bytes, or a callable returning them) -> (payload, crc, size, method).  Members are read back whole either way; deflating
    a few hundred bytes costs more than it saves (zlib set-up per member).  strategy: zlib's (SEQ_STRATEGY for the .seq store)."""
    if callable(data):
        data = data()               # a member whose bytes are made here, off the caller's thread (C emitters: no GIL held)
    if len(data) < 4096:
        return data, zlib.crc32(data), len(data), zipfile.ZIP_STORED
    if strategy == zlib.Z_HUFFMAN_ONLY or strategy == FAST_DEFLATE:
        # the library's own coders, stream and CRC from one call (pep_pack_member): literals only - the sizes of zlib's Z_HUFFMAN_ONLY at several times its
        # rate -, or the single-probe matcher - the sizes of zlib's level 1 at three times its rate; the CRC by carry-less multiplication (zlib's: 1 ms per 2 MB)
        from ._native import pack_member
        payload, crc = pack_member(data, 0 if strategy == zlib.Z_HUFFMAN_ONLY else 1)
        return payload, crc, len(data), zipfile.ZIP_DEFLATED
    from ._native import crc32
    co = zlib.compressobj(1, zlib.DEFLATED, -15, 8, strategy)
    return co.compress(data) + co.flush(), crc32(data), len(data), zipfile.ZIP_DEFLATED


def _archive(where, mode):
    """a store's zip archive: deflate level 1, zip64 when it grows that far"""
    return zipfile.ZipFile(where, mode=mode, compression=zipfile.ZIP_DEFLATED, allowZip64=True, compresslevel=1)


_FAST_APPEND = None


def _fast_append_ok():
    """The fast way below writes through zipfile's private fields (_lock, _writecheck, _didModify, start_dir, filelist, NameToInfo).  A
    CPython that changed them would damage archives silently, so the first use in a process makes one archive in memory that way - a stored
    and a deflated member -, reopens it and has zipfile check it; if anything is off every member goes through ZipFile.writestr instead."""
    global _FAST_APPEND
    if _FAST_APPEND is None:
        try:
            buf, big = io.BytesIO(), bytes(range(256)) * 40
            with _archive(buf, 'w') as zf:
                _append_member(zf, 'a', _pack_member(b'abc'), checked=True)
                _append_member(zf, 'b', _pack_member(big), checked=True)
                zf.writestr('c', b'xyz')
            with zipfile.ZipFile(io.BytesIO(buf.getvalue())) as zf:
                _FAST_APPEND = zf.testzip() is None and zf.namelist() == ['a', 'b', 'c'] and zf.read('a') == b'abc' and zf.read('b') == big and zf.read('c') == b'xyz'
        except Exception:
            _FAST_APPEND = False
    return _FAST_APPEND


def _append_member(zf, name, packed, checked=False):
    """one finished member into an archive that is open for writing: what ZipFile.writestr does after its own compression"""
    payload, crc, size, method = packed
    if not checked and not _fast_append_ok():
        zf.writestr(name, bytes(payload) if method == zipfile.ZIP_STORED else zlib.decompress(bytes(payload), -15))
        return
    zi = zipfile.ZipInfo(name, date_time=time.localtime(time.time())[:6])
    zi.compress_type, zi.external_attr = method, 0o600 << 16
    zi.CRC, zi.compress_size, zi.file_size = crc, len(payload), size
    with zf._lock:
        zf._writecheck(zi)
        zf._didModify = True
        zf.fp.seek(zf.start_dir)
        zi.header_offset = zf.fp.tell()
        zf.fp.write(zi.FileHeader(zip64=None))
        zf.fp.write(payload)
        zf.start_dir = zf.fp.tell()
        zf.filelist.append(zi)
        zf.NameToInfo[zi.filename] = zi


class _LazyZip(object):
    """an archive on disk that is opened for appending (zipfile reads its whole directory then) only when something is asked of it"""

    def __init__(self, fname):
        self.__dict__['_fname'], self.__dict__['_zf'] = fname, None

    def _open(self):
        if self._zf is None:
            self.__dict__['_zf'] = _archive(self._fname, 'a')
        return self._zf

    def __getattr__(self, name):
        return getattr(self._open(), name)

    def __setattr__(self, name, value):
        setattr(self._open(), name, value)

    def close(self):
        if self._zf is not None:
            self._zf.close()


class MapBsn(object):
    """dict-like store: zip member `str(key)` holds one array in .npy format (object arrays pickled).  Readable by
    `np.load(fname, allow_pickle=True)` like the reference's files (PEPPAN.py:1931)."""

    def __init__(self, fname, mode='r'):
        self.fname, self.mode = fname, mode
        self.conn = _archive(fname, mode)
        self.namelist = set(self.conn.namelist())
        # writes go through ONE background thread per store, in order; the members it appends are made and deflated by a shared pool
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
                db, key, packed = item
                if self._error is None:
                    _append_member(db, key, packed if isinstance(packed, tuple) else packed.result())
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
        with zipfile.ZipFile(self.fname) as src, _archive(tmp, 'w') as dst:
            for name in src.namelist():
                if name != key:
                    dst.writestr(name, src.read(name))
        os.replace(tmp, self.fname)
        self.conn = _archive(self.fname, 'a')

    def _start_writer(self):
        if self._thread is None:
            self._queue = queue.Queue(maxsize=256)
            self._thread = threading.Thread(target=self._writer, daemon=True)
            self._thread.start()

    def _enqueue(self, db, key, data, strategy=zlib.Z_DEFAULT_STRATEGY):
        """data: the member's bytes, or a callable that returns them (run by the writer thread)"""
        self._start_writer()
        if not callable(data) and len(data) < 4096:
            packed = _pack_member(data)             # a few hundred bytes, stored as they are: not worth a trip through the pool
        else:
            packed = _packers().submit(_pack_member, data, strategy)
        self._queue.put((db, key, packed))

    def _save(self, db, key, val):
        self._enqueue(db, key, _npy_bytes(val))     # serialised here: the caller may change `val` afterwards

    def save(self, key, val):
        key = str(key)
        self.delete_real(key)
        self._save(self.conn, key, val)
        self.namelist.add(key)

    def save_member(self, key, make_bytes, strategy=zlib.Z_DEFAULT_STRATEGY):
        """save() for a member that arrives as ready-made .npy bytes (or a callable producing them on the writer thread)"""
        key = str(key)
        self.delete_real(key)
        self._enqueue(self.conn, key, make_bytes, strategy)
        self.namelist.add(key)

    def save_packed(self, key, packed):
        """save() for a member that is finished already: (payload, crc, size, method) as _pack_member returns it (made by a worker process)"""
        key = str(key)
        self.delete_real(key)
        self._start_writer()
        self._queue.put((self.conn, key, tuple(packed)))
        self.namelist.add(key)

    def _swap_in(self, members):
        """the archive rebuilt from (key, array) pairs beside the old one, then put in its place (zip members cannot grow where they are)"""
        side = self.fname[:-4] + '.tmp.npz'
        listed = set()
        with _archive(side, 'w') as fresh:
            for key, val in members:
                self._save(fresh, key, val)
                listed.add(key)
            self._flush()
        self.conn.close()
        os.replace(side, self.fname)
        self.namelist = listed
        self.conn = _archive(self.fname, 'a')

    def update(self, dataset):
        """Rows for many keys at once (PEPPAN.py:91-113): every 2-D array of `dataset` belongs to the key in its first cell and is
        appended to what the store holds under that key; members that are not mentioned stay, empty ones are dropped."""
        incoming = {str(rows[0][0]): rows for rows in dataset}         # (a key named twice: the later array counts)
        self._flush()

        def grown():
            for key, rows in incoming.items():
                have = self.get(key)
                yield key, np.vstack([have, rows]) if len(have) else rows
            for key in sorted(self.namelist.difference(incoming)):
                yield key, self.get(key)
        self._swap_in((key, val) for key, val in grown() if len(val))

    def update_table(self, tab, order=None):
        """update() for an int64 table sorted by its first column - or sorted by it when taken in the order `order` -: the rows of every key
        appended to what the store holds under it.  Into an EMPTY store - the usual case, the table being kept in memory until the end - all
        members are made as finished zip entries by host threads of the library (pep_store_tab_members: they gather the rows, too) and
        written with one write(); 10 000 members one by one cost a second"""
        tab = np.ascontiguousarray(tab, dtype=np.int64)
        if len(tab) == 0:
            return
        ids = tab[:, 0] if order is None else tab[:, 0][order]
        starts = np.concatenate([[0], np.flatnonzero(np.diff(ids)) + 1, [len(tab)]]).astype(np.int64)
        if self.namelist or self.conn.filelist or self.mode == 'r' or not _fast_append_ok():
            if order is not None:
                tab = tab[order]
            return self.update([tab[a:b] for a, b in zip(starts[:-1].tolist(), starts[1:].tolist())])
        from . import _native
        self._flush()
        keys = ids[starts[:-1]]
        stamp = time.localtime(time.time())[:6]
        whole = _native.store_tab_archive(tab, starts, keys, stamp, order=order) if self.mode == 'w' else None
        if whole is not None:
            # the complete archive, directory included, from the library: no zipfile object is made for it unless the store is used again
            self.conn.close()
            with open(self.fname, 'wb') as f:
                f.write(memoryview(whole))
            self.conn = _LazyZip(self.fname)
            self.namelist = set(map(str, keys.tolist()))
            return
        blob, crc, csize, usize, at = _native.store_tab_members(tab, starts, keys, stamp, order=order)
        zf = self.conn
        with zf._lock:
            zf._didModify = True
            zf.fp.seek(zf.start_dir)
            base = zf.fp.tell()
            zf.fp.write(memoryview(blob))
            zf.start_dir = zf.fp.tell()
            for name, c, cs, us, off in zip(map(str, keys.tolist()), crc.tolist(), csize.tolist(), usize.tolist(), at.tolist()):
                zi = zipfile.ZipInfo(name, date_time=stamp)
                zi.compress_type, zi.external_attr = zipfile.ZIP_DEFLATED if us >= 4096 else zipfile.ZIP_STORED, 0o600 << 16      # (the library's rule, csrc/stores.hip tab_members: members of 4 KiB and more are deflated - whatever size came out)
                zi.CRC, zi.compress_size, zi.file_size, zi.header_offset = c, cs, us, base + off
                zf.filelist.append(zi)
                zf.NameToInfo[name] = zi
                self.namelist.add(name)


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


def _known_fraction(T, old_prediction):
    """Column 10 of the mapping tables (PEPPAN.py:869-901): for every hit the largest fraction of an original gene of the same contig
    that it covers in frame and on the same strand (0.1 when there is none), for all rows of the HitTable at once.  Returns
    (order, known): `order` = the row order in which the reference walks the table, (contig, lower reference coordinate), stable;
    `known[k]` belongs to row order[k].
    An original gene p = [id, start, end, strand, ...] counts for a hit [s, e] when one of its two frame markers is one of the hit's
    and the overlap is >= 0.6 of the gene or of the hit; the hit's markers come from where its query would start / end on the
    reference.  The reference sweeps the contig's genes with a pointer that only moves forward past genes ending before the hit
    and stops at the first gene starting behind it; with the genes in start order (as the store holds them) that is "every
    overlapping gene", which is what the vectorised form evaluates - an unsorted gene list takes the row loop instead."""
    n = len(T)
    lo, hi = np.minimum(T.ss, T.se), np.maximum(T.ss, T.se)
    order = np.lexsort((lo, T.r_codes()))
    known = np.full(n, 0.1, dtype=np.float64)
    if n == 0:
        return order, known
    ri = T.ri[order]
    s, e = lo[order], hi[order]
    fwd = (T.ss < T.se)[order]
    head, tail = (T.ss - T.qs + 1)[order], (T.se + (T.ql - T.qe))[order]
    f1 = np.where(fwd, head % 3 + 1, (-head) % 3 - 1)
    f2 = np.where(fwd, (tail + 1) % 3 + 1, (-(tail - 1)) % 3 - 1)
    bounds = np.concatenate([[0], np.flatnonzero(np.diff(ri)) + 1, [n]])
    with (contextlib.nullcontext(old_prediction) if isinstance(old_prediction, MapBsn) else MapBsn(old_prediction)) as op:       # (an open store is read as it is)
        for a, b in zip(bounds[:-1].tolist(), bounds[1:].tolist()):
            genes = op.get(T.r_tab[ri[a]])
            if len(genes) == 0:
                continue
            if isinstance(genes, np.ndarray) and genes.ndim == 2 and genes.shape[1] >= 4:       # (the store's object rows: three column conversions in C instead of three loops)
                g1, g2 = genes[:, 1].astype(np.int64), genes[:, 2].astype(np.int64)
                plus = np.asarray(genes[:, 3] == '+', dtype=bool)
            else:
                g1 = np.array([p[1] for p in genes], dtype=np.int64)
                g2 = np.array([p[2] for p in genes], dtype=np.int64)
                plus = np.array([p[3] == '+' for p in genes], dtype=bool)
            if np.any(np.diff(g1) < 0):
                known[a:b] = _known_fraction_rows(s[a:b], e[a:b], f1[a:b], f2[a:b], g1, g2, plus)
                continue
            # genes [first, last) can touch the hit: first = the first gene (in store order) that does not end before it
            first = np.searchsorted(np.maximum.accumulate(g2), s[a:b], side='left')
            last = np.maximum(np.searchsorted(g1, e[a:b], side='right'), first)
            cnt = last - first
            tot = int(cnt.sum())
            if tot == 0:
                continue
            row = np.repeat(np.arange(a, b), cnt)
            gi = np.repeat(first - np.concatenate([[0], np.cumsum(cnt)[:-1]]), cnt) + np.arange(tot)
            p1, p2, pl = g1[gi], g2[gi], plus[gi]
            m1 = np.where(pl, p1 % 3 + 1, (-(p1 - 1)) % 3 - 1)
            m2 = np.where(pl, (p2 + 1) % 3 + 1, (-p2) % 3 - 1)
            in_frame = (m1 == f1[row]) | (m1 == f2[row]) | (m2 == f1[row]) | (m2 == f2[row])
            plen = p2 - p1 + 1
            ovl = (np.minimum(e[row], p2) - np.maximum(s[row], p1) + 1).astype(np.float64)
            ok = in_frame & ((ovl >= 0.6 * plen) | (ovl >= 0.6 * (e[row] - s[row] + 1)))
            if ok.any():
                np.maximum.at(known, row[ok], ovl[ok] / plen[ok])
    return order, known


def _known_fraction_rows(s, e, f1, f2, g1, g2, plus):
    """the reference's pointer sweep, row by row (genes not in start order)"""
    out = np.full(len(s), 0.1, dtype=np.float64)
    at, ng = 0, len(g1)
    for k in range(len(s)):
        while at < ng and s[k] > g2[at]:
            at += 1
        for j in range(at, ng):
            if e[k] < g1[j]:
                break
            if plus[j]:
                m1, m2 = g1[j] % 3 + 1, (g2[j] + 1) % 3 + 1
            else:
                m1, m2 = (-(g1[j] - 1)) % 3 - 1, (-g2[j]) % 3 - 1
            if m1 not in (f1[k], f2[k]) and m2 not in (f1[k], f2[k]):
                continue
            plen = g2[j] - g1[j] + 1
            ovl = min(e[k], g2[j]) - max(s[k], g1[j]) + 1.
            if ovl >= 0.6 * plen or ovl >= 0.6 * (e[k] - s[k] + 1):
                out[k] = max(out[k], ovl / plen)
    return out


def _with_known(T, old_prediction):
    """the table in the order compare_prediction returns it - (query, contig, score), stable on top of the (contig, position) walk -
    with column 10 replaced.  One pass of host C++ over the columns (pep_known_order: both orders and the interval sweep; round 6 - as numpy expressions,
    _with_known_numpy below, this was 3.2 of the 5.4 ms build_groups spent on a genome); the store is asked for the genes of the contigs the table names"""
    from . import _native as N
    if len(T) == 0:
        return T
    with (contextlib.nullcontext(old_prediction) if isinstance(old_prediction, MapBsn) else MapBsn(old_prediction)) as op:       # (an open store is read as it is)
        def genes_of(c):
            genes = op.get(T.r_tab[c])
            if len(genes) == 0:
                return None
            if isinstance(genes, np.ndarray) and genes.ndim == 2 and genes.shape[1] >= 4:       # (the store's object rows: three column conversions in C instead of three loops)
                return genes[:, 1].astype(np.int64), genes[:, 2].astype(np.int64), np.asarray(genes[:, 3] == '+', dtype=bool)
            return (np.array([p[1] for p in genes], dtype=np.int64), np.array([p[2] for p in genes], dtype=np.int64), np.array([p[3] == '+' for p in genes], dtype=bool))
        order, known = N.known_order(T, genes_of)
    T = T.take(order)
    T.evalue = known
    return T


def _with_known_numpy(T, old_prediction):
    """_with_known as numpy expressions over the columns (_known_fraction + two lexsorts): the statement pep_known_order is held to (tests/test_mapbsn_golden.py)"""
    order, known = _known_fraction(T, old_prediction)
    # (the second sort is stable on top of the first order: sorted over the three key columns taken in that order, the table itself is gathered once)
    again = np.lexsort((T.score[order], T.r_codes()[order], T.q_codes()[order]))
    T = T.take(order[again])
    T.evalue = known[again]
    return T


def compare_prediction(blastab, old_prediction):
    """column 10 <- the largest fraction of an in-frame, same-strand original gene that a hit covers (0.1 if none);
    returns the table sorted by (query, contig, score).  PEPPAN.py:869-901.  Object rows in and out."""
    from .hittable import HitTable
    if blastab.shape[0] == 0:
        return blastab
    return _with_known(HitTable.from_rows(blastab), old_prediction).to_rows(cigar='str')


# ------------------------------------------------------------------------------------------------ one genome
def _passes_all(length, ql, params):
    return ((length >= np.maximum(params['match_prop'] * ql, params['match_len'])) |
            (length >= np.maximum(params['match_prop1'] * ql, params['match_len1'])) |
            (length >= np.maximum(params['match_prop2'] * ql, params['match_len2'])))


class GenomeGroups(object):
    """The groups of merged hits of ONE genome as columns: what the reference's `bsn` table holds per row - [gene, contig, score,
    identity, packed allele, group id, hit rows] (PEPPAN.py:836-866) - before any Python object is made for it.
      gene / contig   int64[n]    names of the group's leading hit
      score / iden    float64[n]
      packed, pack_off            the base-5 packed alleles back to back, group g = packed[pack_off[g]:pack_off[g+1]]
      rows, row_off               HitTable of the hit rows of all groups, group g = rows [row_off[g], row_off[g+1])
      ovl             int64[k, 3] pairs of groups (local ids) that overlap on the genome, with their relation class"""
    __slots__ = ('gene', 'contig', 'score', 'iden', 'packed', 'pack_off', 'rows', 'row_off', 'ovl')

    def __init__(self, gene, contig, score, iden, packed, pack_off, rows, row_off, ovl):
        self.gene, self.contig, self.score, self.iden = gene, contig, score, iden
        self.packed, self.pack_off, self.rows, self.row_off, self.ovl = packed, pack_off, rows, row_off, ovl

    def __len__(self):
        return len(self.gene)

    @classmethod
    def none(cls):
        z = np.zeros(0, dtype=np.int64)
        return cls(z, z, np.zeros(0), np.zeros(0), np.zeros(0, np.uint8), np.zeros(1, np.int64), None, np.zeros(1, np.int64), np.zeros([0, 3], dtype=np.int64))

    def as_bsn(self):
        """the reference's object table [n, 7] (what iter_map_bsn stores in <prefix>.<id>.bsn.npz, PEPPAN.py:866)"""
        n = len(self)
        bsn = np.empty([n, 7], dtype=object)
        if n == 0:
            return bsn
        rows16 = self.rows.to_rows(cigar='str')
        gene, contig, iden = self.gene.tolist(), self.contig.tolist(), self.iden.tolist()
        r_lo, p_lo = self.row_off.tolist(), self.pack_off.tolist()
        for g in range(n):
            row = bsn[g]
            row[0], row[1], row[2], row[3], row[4], row[5], row[6] = gene[g], contig[g], self.score[g], iden[g], self.packed[p_lo[g]:p_lo[g + 1]], g, rows16[r_lo[g]:r_lo[g + 1]]
        return bsn


class OrthoRelation(object):
    """the relation of two genes in the all-vs-all result (rows [gene, gene, value] of <prefix>.self_bsn.npy): +1 ortholog-like
    (value > 0), -1 conflict (value < 0), 2 when the pair is not listed; a gene with itself is 0.  The reference builds a dictionary
    of both orientations for every genome (PEPPAN.py:856-862: first every pair as listed, then every pair reversed, later entries
    replacing earlier ones); here it is ONE sorted key array per run and a binary search per question."""

    def __init__(self, ortho):
        og = np.load(ortho, allow_pickle=True) if isinstance(ortho, str) else np.asarray(ortho)
        og = og[og.T[2] != 0] if len(og) else np.zeros([0, 3], dtype=np.int64)
        a, b = og.T[0].astype(np.int64), og.T[1].astype(np.int64)
        sign = np.where(og.T[2].astype(np.int64) > 0, 1, -1)
        self._wide = len(og) > 0 and (min(a.min(), b.min()) < 0 or max(a.max(), b.max()) >= (1 << 31))
        if self._wide:                                           # (names beyond 31 bits: PEPPAN's encoded ids never are)
            self._table = {}
            for x, y, v in list(zip(a.tolist(), b.tolist(), sign.tolist())) + list(zip(b.tolist(), a.tolist(), sign.tolist())):
                self._table[(x, y)] = v
            return
        key = np.concatenate([(a << 32) | b, (b << 32) | a])
        val = np.concatenate([sign, sign])
        order = np.argsort(key, kind='stable')
        key, val = key[order], val[order]
        last = np.concatenate([key[1:] != key[:-1], [True]]) if len(key) else np.zeros(0, dtype=bool)        # of equal keys the latest entry counts
        self._key, self._val = key[last], val[last]

    def between(self, m, k):
        m, k = np.asarray(m, dtype=np.int64), np.asarray(k, dtype=np.int64)
        if self._wide:
            return np.array([0 if x == y else self._table.get((x, y), 2) for x, y in zip(m.tolist(), k.tolist())], dtype=np.int64)
        out = np.full(len(m), 2, dtype=np.int64)
        if len(self._key) and len(m):
            narrow = (m >= 0) & (k >= 0) & (m < (1 << 31)) & (k < (1 << 31))
            q = (m << 32) | k
            at = np.minimum(np.searchsorted(self._key, q), len(self._key) - 1)
            hit = narrow & (self._key[at] == q)
            out[hit] = self._val[at[hit]]
        out[m == k] = 0
        return out


_INT_NAMES = {}


def _int_names(tab):
    """a name table as a list of Python ints, remembered per table object: the genomes of a batch share their tables (10 000 gene ids), and every
    sort of the chain asks for the numeric form"""
    from .hittable import int_name_array
    hit = _INT_NAMES.get(id(tab))
    if hit is None or hit[0] is not tab:
        if len(_INT_NAMES) >= 16:
            _INT_NAMES.clear()
        ints = [int(x) for x in tab]
        int_name_array(ints, register=np.asarray(ints, dtype=np.int64))
        hit = _INT_NAMES[id(tab)] = (tab, ints)
    return hit[1]


def build_groups(blastab, overlap, seq, ortho, old_prediction, params, ctx=None):
    """(17-column table with merge groups, int[m, 3] overlaps) of ONE genome -> GenomeGroups (PEPPAN.py:773-866; the steps: _build_groups_steps)"""
    return build_groups_round([(blastab, overlap, seq)], ortho, old_prediction, params, ctx)[0]


def build_groups_round(items, ortho, old_prediction, params, ctx=None):
    """build_groups for several genomes - items = [(blastab, overlap, seq), ...] - with ONE K12 call for all of them: a genome's K12 is 0.1 ms of kernels, and
    on a GPU that eight worker processes share each call waits milliseconds for its turn (4 ms per genome in a worker's groups thread).  The genomes' requests
    are put behind one another - contig, group and CIGAR indices shifted - and the answer is cut back per genome; the result is what one call per genome gives."""
    steps = [_build_groups_steps(blastab, overlap, seq, ortho, old_prediction, params) for blastab, overlap, seq in items]
    out, asked = [None] * len(steps), []
    for i, g in enumerate(steps):
        try:
            asked.append((i, next(g)))
        except StopIteration as e:                  # (a genome without groups never asks)
            out[i] = e.value
    if asked:
        if ctx is None:
            from .uberBlast import get_context
            ctx = get_context()
        if len(asked) == 1:
            answers = [ctx.alleles(*asked[0][1])]
        else:
            from ._native import LOCUS_DTYPE
            same_arena = all(r[2] is asked[0][1][2] for _, r in asked)
            contigs, loci, arenas, grp_off, qlen, n_rows, n_pack = [], [], [], [], [], [], []
            base_c = base_g = base_r = base_a = 0
            for _, (cs, lc, arena, go, ql, gtable) in asked:
                lc = np.array(lc, dtype=LOCUS_DTYPE)
                lc['contig'] += base_c
                lc['group'] += base_g
                if not same_arena:
                    lc['cigar_off'] += base_a
                    arenas.append(np.asarray(arena, dtype=np.uint32))
                    base_a += len(arena)
                contigs += list(cs)
                loci.append(lc)
                go = np.asarray(go, dtype=np.int64)
                grp_off.append(go[:-1] + base_r)
                qlen.append(np.asarray(ql, dtype=np.int64))
                n_rows.append(len(lc))
                n_pack.append(int(((qlen[-1] + 2) // 3).sum()))
                base_c, base_g, base_r = base_c + len(cs), base_g + len(ql), base_r + len(lc)
            in_frame, orf, packed = ctx.alleles(contigs, np.concatenate(loci), asked[0][1][2] if same_arena else np.concatenate(arenas),
                                                np.concatenate(grp_off + [np.array([base_r], dtype=np.int64)]).astype(np.uint64), np.concatenate(qlen), asked[0][1][5])
            answers, r0, p0 = [], 0, 0
            for nr, npk in zip(n_rows, n_pack):
                answers.append((in_frame[r0:r0 + nr], orf[r0:r0 + nr], packed[p0:p0 + npk]))
                r0, p0 = r0 + nr, p0 + npk
        for (i, _), answer in zip(asked, answers):
            try:
                steps[i].send(answer)
                raise RuntimeError('build_groups: the steps of a genome asked twice')
            except StopIteration as e:
                out[i] = e.value
    return out


def _build_groups_steps(blastab, overlap, seq, ortho, old_prediction, params):
    """The steps of build_groups for one genome as a generator: it yields ONE request - the arguments of K12 (`Context.alleles`) - is sent the answer,
    and returns the GenomeGroups (a genome without groups returns at once).  (17-column table with merge groups, int[m, 3] overlaps) -> GenomeGroups.  PEPPAN.py:773-866.
    `blastab` is the HitTable the search chain ends with (the product path: no Python row is made at all) or the same thing as object
    rows; `ortho` an OrthoRelation (or what it is built from).  The per-hit allele strings, their in-frame / stop-free lengths and the
    packing run on the GPU (K12, `ctx.alleles`).

    Groups, in the reference's order: first every row that stands for itself - a row whose merge group is just itself, or a member
    of a chain that also passes the thresholds alone - in table order; then the chains, in the order their first member shows up,
    members in chain order."""
    from .hittable import HitTable
    T = blastab if isinstance(blastab, HitTable) else HitTable.from_rows(blastab)
    if len(T) == 0:
        return GenomeGroups.none()
    T.q_tab, T.r_tab = _int_names(T.q_tab), _int_names(T.r_tab)
    T.q_sorted = T.r_sorted = False           # (integer names order numerically from here on)
    T = _with_known(T, old_prediction)
    n = len(T)
    n_id = int(T.rid.max()) + 1
    mi = params['match_identity']
    ok = (T.m_span >= 0) & (T.m_iden >= mi) & _passes_all(T.m_span, T.ql, params)         # the row's merge group passes
    kept = np.zeros(n_id, dtype=bool)
    kept[T.rid[ok]] = True
    lone = ok & (T.m_len <= 1)
    also_alone = ok & (T.m_len > 1) & (T.iden >= mi) & _passes_all(T.qe - T.qs + 1, T.ql, params)
    single = np.flatnonzero(lone | also_alone)
    # chains: keyed by their first member's row id, in order of first appearance; every member row finds its slot by its row id
    in_chain = np.flatnonzero(ok & (T.m_len > 1))
    chain_rows, chain_first = [], []
    if len(in_chain):
        key = T.m_ids[T.m_start[in_chain]]
        uniq, first_pos = np.unique(key, return_index=True)
        row_of_id = np.full(n_id, -1, dtype=np.int64)
        row_of_id[T.rid[in_chain]] = in_chain
        rep = in_chain[np.sort(first_pos)]                                         # one member row per chain, same order
        for r in rep.tolist():
            members = T.m_ids[T.m_start[r]:T.m_start[r] + T.m_len[r]]
            rows = row_of_id[members]
            if (rows < 0).any():
                raise ValueError('build_bsn: a chained hit is missing from the table')
            chain_rows.append(rows)
            chain_first.append(r)
    flat = np.concatenate([single] + chain_rows).astype(np.int64) if (len(single) or chain_rows) else np.zeros(0, np.int64)
    n_rows = np.concatenate([np.ones(len(single), dtype=np.int64), np.array([len(r) for r in chain_rows], dtype=np.int64)])
    n_groups = len(n_rows)
    overlap = overlap[kept[overlap.T[0]] & kept[overlap.T[1]], :2]
    if n_groups == 0:
        return GenomeGroups.none()
    grp_off = np.concatenate([[0], np.cumsum(n_rows)]).astype(np.uint64)
    first = grp_off[:-1].astype(np.int64)
    head_row = np.concatenate([single, np.array(chain_first, dtype=np.int64)]).astype(np.int64)      # the row a group takes gene / contig / group score from
    is_lone_group = np.concatenate([lone[single], np.zeros(len(chain_rows), dtype=bool)])
    # ---- K12 over every row of every group
    from ._native import LOCUS_DTYPE
    contigs = [sq for nm, sq in seq]
    cidx = {nm: i for i, (nm, sq) in enumerate(seq)}
    loci = np.zeros(len(flat), dtype=LOCUS_DTYPE)
    contig_of = np.array([cidx.get(nm, -1) for nm in T.r_tab], dtype=np.int64)[T.ri[flat]]       # (the name table of a batch search lists every genome's contigs)
    if len(contig_of) and contig_of.min() < 0:
        raise ValueError('build_bsn: a hit names a contig that is not part of this genome')
    loci['contig'] = contig_of
    loci['q_start'], loci['rs'], loci['re'] = T.qs[flat], T.ss[flat], T.se[flat]
    loci['cigar_runs'], loci['cigar_off'] = T.c_runs[flat], T.c_off[flat]
    loci['group'] = np.repeat(np.arange(n_groups), n_rows)
    ql = T.ql[flat]
    in_frame, orf, packed = yield (contigs, loci, T.arena, grp_off, ql[first], params['gtable'])
    sc = np.minimum(in_frame, orf + 3)
    iden, known = T.iden[flat], T.evalue[flat]
    q_lo, q_hi = T.qs[flat], T.qe[flat]
    qspan = q_hi - q_lo + 1
    r = np.sqrt(sc.astype(np.float64) / ql * known)
    msc = (sc * iden) * np.sqrt(sc * r)
    amsc = msc / qspan
    pack_off = np.concatenate([[0], np.cumsum((ql[first] + 2) // 3)]).astype(np.int64)
    # ---- group scores: a single row's own; the fragments of a chain that overlap on the query give way to the stronger neighbour first
    score = msc[first].astype(np.float64)
    row_off = grp_off.astype(np.int64)
    for gid in np.flatnonzero(n_rows > 1).tolist():
        lo, hi = int(row_off[gid]), int(row_off[gid + 1])
        spans = [[q_lo[k], q_hi[k], amsc[k], msc[k]] for k in range(lo, hi)]
        for prev, cur in zip(spans[:-1], spans[1:]):
            if cur[0] < prev[1]:
                if cur[2] > prev[2]:
                    prev[1] = cur[0] - 1
                    prev[3] = prev[2] * (prev[1] - prev[0] + 1)
                else:
                    cur[0] = prev[1] + 1
                    cur[3] = cur[2] * (cur[1] - cur[0] + 1)
        score[gid] = np.sum([c[3] for c in spans])
    # group identity: the merge group's for a lone row and for a chain, the row's own for a chain member standing alone
    g_iden = np.where(is_lone_group | (n_rows > 1), T.m_iden[head_row], T.iden[head_row])
    from .hittable import int_name_array
    q_names, r_names = int_name_array(T.q_tab), np.asarray(T.r_tab, dtype=np.int64)
    g_gene, g_contig = q_names[T.qi[head_row]], r_names[T.ri[head_row]]
    # ---- overlaps between groups: a row id may stand in a group of its own and in a chain
    as_single, as_chain = np.full(n_id, -1, dtype=np.int64), np.full(n_id, -1, dtype=np.int64)
    gid_of_row = loci['group'].astype(np.int64)
    multi = np.repeat(n_rows > 1, n_rows)
    rid_flat = T.rid[flat]
    as_single[rid_flat[~multi]] = gid_of_row[~multi]
    as_chain[rid_flat[multi]] = gid_of_row[multi]
    a0, c0, a1, c1 = as_single[overlap.T[0]], as_chain[overlap.T[0]], as_single[overlap.T[1]], as_chain[overlap.T[1]]
    pairs = np.vstack([np.vstack([m, k]).T[(m >= 0) & (k >= 0)] for m in (a0, c0) for k in (a1, c1)] +
                      [np.vstack([as_single, as_chain]).T[(as_single >= 0) & (as_chain >= 0)]])
    if pairs.shape[0]:
        rel = ortho if isinstance(ortho, OrthoRelation) else OrthoRelation(ortho)
        cls = rel.between(g_gene[pairs.T[0]], g_gene[pairs.T[1]])
        pairs = np.hstack([pairs, cls[:, np.newaxis]])[cls >= 0]
    else:
        pairs = np.zeros([0, 3], dtype=np.int64)
    rows = T.take(flat)
    rows.merge = rows.m_span = None                     # (the stored rows are the 16 columns: no merge lists to build)
    return GenomeGroups(g_gene, g_contig, score, g_iden, packed, pack_off, rows, row_off, pairs)


def build_bsn(blastab, overlap, seq, orthoGroup, old_prediction, params, ctx=None):
    """build_groups in the reference's form: (bsn object[n, 7], ovl int[k, 3]) with bsn row = [gene, contig, score, identity,
    packed allele, group id, rows(object[k, 16])] (PEPPAN.py:773-866)"""
    G = build_groups(blastab, overlap, seq, orthoGroup, old_prediction, params, ctx)
    return G.as_bsn(), G.ovl


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
ONE_PROCESS_BATCH = 8                  # genomes per search of a mapping without worker processes (the next batch is searched while this one gets its groups)
POOL_ROUND = 8                         # genomes per round of a worker pool on one GPU (get_map_bsn) ...
POOL_ROUND_NT = 32000000               # ... and nucleotides per round (mean genome size x genomes)
MAT_STRATEGY = FAST_DEFLATE            # how the members of the .mat store are deflated: hit rows as a pickle stream - repeated opcodes, similar numbers; zlib's level 1 took 13 ms of CPU per mapped genome
SEQ_STRATEGY = zlib.Z_HUFFMAN_ONLY      # how the members of the .seq store are deflated.  Packed alleles are all but incompressible by matching (a byte
#                       holds three bases of three different thirds of an allele): entropy coding alone (pep_deflate_literals, 590 MB/s; zlib's Z_HUFFMAN_ONLY: 130) makes 0.75 of a gene set's alleles where
#                       level 1's match search makes 0.77 at 30 MB/s, and those 50 ms of CPU per genome were two thirds of what a mapped genome costs on the
#                       host (16 CPUs granted to a GPU box: 148 -> 197 genomes/s with eight workers).  Where a genome's groups repeat a locus - once per
#                       paralogous exemplar it matches, identical spans only - matching does find something: 0.66 against 0.81 on the synthetic genomes, whose
#                       family members all have the same length.  zlib.Z_DEFAULT_STRATEGY brings that back.
CHUNK = 1000          # arrays per member of the .seq / .mat stores (PEPPAN.py:953, 962)
BLOCK = 30000         # group ids per member of the .conflicts store (PEPPAN.py:934-947)
TABLE_ROWS = 48 << 20 # gene-table rows kept in memory between two updates of the .tab store (7 x int64 each: 2.7 GB; 2 000 genomes x 6 600 groups are 13 M rows).  The
#                       reference updates every 500 genomes (PEPPAN.py:972) to bound the memory of ITS rows - Python objects; an update of a store that holds
#                       something already rewrites the whole archive member by member (5 s for 10 000 genes: measured when the limit was 8 M rows and a
#                       2 000-genome run crossed it), the first one is a single call into the library (0.2 - 0.8 s), and the store's content does not depend
#                       on how often that happens


def _gpu_search(prefix, clust, jobs, params, genomes_per_batch=64):
    """yield (blastab, overlap) per genome, `genomes_per_batch` genomes per GPU search.  The genomes go to the search as they are -
    the reference writes every genome to <prefix>.<id>.genome for its uberBlast call to read back (PEPPAN.py:763-766)"""
    from .uberBlast import uberBlastBatch
    argv = _map_argv(clust, params)
    for lo in range(0, len(jobs), genomes_per_batch):
        # a contig handed over as str is turned into ASCII bytes ONCE, in place in the job's own list: the search packs it, K12 (build_groups) packs it again, and
        # every packing of a str is another encode of the genome (1.2 ms per genome; the pool's workers get bytes from their files anyway)
        for id, taxon, seq in jobs[lo:lo + genomes_per_batch]:
            for pair in seq:
                if isinstance(pair, list) and isinstance(pair[1], str):
                    try:
                        pair[1] = pair[1].encode('ascii')
                    except UnicodeEncodeError:
                        pass
        # (strict: a search tool that fails fails the mapping - stores made of one tool's hits only would look like results)
        for r in uberBlastBatch([seq for id, taxon, seq in jobs[lo:lo + genomes_per_batch]], argv, as_tables=True, strict=True):        # (HitTable, overlaps) per genome
            yield r


def _ahead(make, depth):
    """the items of the generator make() - made by a thread of its own, at most `depth` of them ahead of the consumer.  The genome mapping of ONE
    process runs its searches through this: a search's waits for the GPU happen inside the library, without the interpreter lock, while the caller's
    thread makes the groups of the genomes in front (the worker processes of mapworkers.py do the same thing with their two threads).  Whatever
    the generator raises is raised here; a consumer that stops early stops the thread at its next item."""
    import queue
    import threading
    box, stop = queue.Queue(maxsize=max(1, depth)), threading.Event()

    def produce():
        try:
            gen = make()
            try:
                for item in gen:
                    while not stop.is_set():
                        try:
                            box.put(('item', item), timeout=0.2)
                            break
                        except queue.Full:
                            pass
                    if stop.is_set():
                        return
            finally:
                gen.close()
            while not stop.is_set():                 # (a consumer that stopped with the queue full must not leave this thread blocked for good)
                try:
                    box.put(('end', None), timeout=0.2)
                    break
                except queue.Full:
                    pass
        except BaseException as e:
            while not stop.is_set():
                try:
                    box.put(('error', e), timeout=0.2)
                    return
                except queue.Full:
                    pass

    t = threading.Thread(target=produce, daemon=True)
    t.start()
    try:
        while True:
            kind, val = box.get()
            if kind == 'item':
                yield val
            elif kind == 'error':
                raise val
            else:
                return
    finally:
        stop.set()
        t.join(30.)


def _dist_world(group):
    dist = sys.modules.get('torch.distributed')            # only a caller that set up a process group has imported it
    if dist is not None and dist.is_available() and dist.is_initialized():
        return dist, dist.get_world_size(group), dist.get_rank(group)
    return None, 1, 0


def _all_groups(prefix, clust, jobs, ortho, old_prediction, params, search, ctx, group, per_round, pool=None):
    """(job, GenomeGroups) for every genome in job order.  With torch.distributed initialised the genomes are dealt to the ranks in
    blocks of `per_round` (independent units, no data-path collective); every rank maps its block on its own GPU and rank 0
    gathers the finished per-genome columns - only rank 0 yields, the others just take part.  With a pool of worker processes
    (mapworkers.MapWorkers) a rank's genomes are dealt to its workers instead of being mapped by the rank itself."""
    def local(mine):
        if pool is not None:
            return [G for job, G in pool.rounds(mine, max(1, min(per_round, -(-len(mine) // pool.n))))] if mine else []
        out = []
        for (id, taxon, seq), (blastab, overlap) in zip(mine, search(prefix, clust, mine, params) if mine else ()):
            out.append(build_groups(blastab, overlap, seq, ortho, old_prediction, params, ctx))
        return out
    dist, world, rank = _dist_world(group)
    if world == 1:
        if pool is not None:
            for job, G in pool.rounds(jobs, per_round):     # (form 'members': the jobs of a round, what the stores take from it)
                yield job, G
            return
        found = search(prefix, clust, jobs, params) if not getattr(search, 'runs_ahead', False) else _ahead(lambda: search(prefix, clust, jobs, params), 2 * per_round)
        for job, (blastab, overlap) in zip(jobs, found):
            yield job, build_groups(blastab, overlap, job[2], ortho, old_prediction, params, ctx)
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
                block = jobs[lo_r:lo_r + per_round]
                if len(gathered[r]) != len(block):          # (a rank that dealt by another unit: stores made of this would lose or repeat genomes)
                    raise RuntimeError('get_map_bsn: rank %d returned %d genomes for a block of %d (genomes_per_round must be the same on every rank)' % (r, len(gathered[r]), len(block)))
                for job, out in zip(block, gathered[r]):
                    yield job, out


class _ConflictBlocks(object):
    """The .conflicts store (PEPPAN.py:934-947, 983-986): for every block of BLOCK consecutive group ids one member holding a CSR -
    BLOCK + 1 offsets (into the member itself, so they start at BLOCK + 1) followed by one value per conflict, partner id * 10 + class -
    listing, per group of the block, the groups it overlaps on its genome, both directions of every pair.
    Conflicts never cross genomes and group ids grow genome by genome, so the entries arrive ordered by block and a block is complete
    as soon as the group counter has passed its end; what is pending is a list of (group id, value) column pairs."""

    def __init__(self, store):
        self.store, self.pending = store, []

    def add_sorted(self, src, val):
        """a genome's conflicts as the doubled list group -> partner * 10 + class, sorted by group, store-wide group ids (StoreBlock)"""
        if len(src):
            self.pending.append((src, val))

    def write(self, below=None):
        """every block that ends at or before group id `below` (None: everything that is left)"""
        if not self.pending:
            return
        src, val = (np.concatenate(c) for c in zip(*self.pending))
        block = src // BLOCK
        done = len(src) if below is None else int(np.searchsorted(block, below // BLOCK))
        if done == 0:
            return
        starts = np.concatenate([[0], np.flatnonzero(np.diff(block[:done])) + 1, [done]])
        for a, b in zip(starts[:-1].tolist(), starts[1:].tolist()):
            per_group = np.bincount(src[a:b] - block[a] * BLOCK, minlength=BLOCK)
            self.store.save(int(block[a]), np.concatenate([np.concatenate([[0], np.cumsum(per_group)]) + (BLOCK + 1), val[a:b]]))
        self.pending = [(src[done:], val[done:])] if done < len(src) else []


class _MemberQueue(object):
    """Groups waiting to become members of CHUNK consecutive groups each (the .seq and .mat stores, PEPPAN.py:950-966): a list of
    per-genome column blocks with the range of groups still unwritten; a member is cut as soon as CHUNK groups are there and handed to
    the store's writer thread as a closure, which emits the .npy bytes from the columns (pep_store_*_member) and deflates them."""

    def __init__(self, store, emit, strategy=zlib.Z_DEFAULT_STRATEGY):
        self.store, self.emit, self.parts, self.waiting, self.members, self.strategy = store, emit, [], 0, 0, strategy

    def add(self, block, n_groups):
        self.parts.append([block, 0, n_groups])
        self.waiting += n_groups
        while self.waiting >= CHUNK:
            self._cut(CHUNK)

    def add_finished(self, first_member, packed):
        """whole members made elsewhere (a worker process): (payload, crc, size, method) each, numbered from `first_member`"""
        if not packed:
            return
        if self.waiting or self.members != first_member:
            raise RuntimeError('store members out of step: %d groups waiting, member %d expected, %d delivered' % (self.waiting, self.members, first_member))
        for p in packed:
            self.store.save_packed(self.members, p)
            self.members += 1

    def close(self):
        if self.waiting:
            self._cut(self.waiting)

    def _cut(self, want):
        take, need = [], want
        while need:
            block, lo, hi = self.parts[0]
            n = min(need, hi - lo)
            take.append((block, lo, lo + n))
            need -= n
            if lo + n == hi:
                self.parts.pop(0)
            else:
                self.parts[0][1] = lo + n
        self.waiting -= want
        emit = self.emit
        self.store.save_member(self.members, lambda: emit(take), self.strategy)
        self.members += 1


def _mat_block(G):
    """the stored columns of a genome's hit rows, CIGAR runs gathered into an arena of their own (the search's arena is shared by a
    whole batch of genomes)"""
    R = G.rows
    runs = R.c_runs
    start = np.concatenate([[0], np.cumsum(runs)]).astype(np.int64)
    src = np.repeat(R.c_off - start[:-1], runs) + np.arange(int(start[-1]))
    from .hittable import int_name_array
    q_num = int_name_array(R.q_tab)
    q, r = (q_num if q_num is not None else np.asarray(R.q_tab, dtype=np.int64))[R.qi], np.asarray(R.r_tab, dtype=np.int64)[R.ri]
    cols = [q, r, R.iden, R.aln, R.mis, R.gap, R.qs, R.qe, R.ss, R.se, R.evalue, R.score, R.ql, R.sl]
    return dict(cols=cols, arena=R.arena[src], run_off=start, rid=R.rid, row_off=G.row_off, score_is_int=R.score_is_int)


def _emit_mat(take):
    from . import _native as N
    cols, arenas, c_off, c_runs, rids, offs = [[] for _ in range(14)], [], [], [], [], [np.zeros(1, np.int64)]
    runs_before = rows_before = 0
    as_int = True
    for block, lo, hi in take:
        a, b = int(block['row_off'][lo]), int(block['row_off'][hi])
        for k, c in enumerate(block['cols']):
            cols[k].append(c[a:b])
        ra, rb = int(block['run_off'][a]), int(block['run_off'][b])
        arenas.append(block['arena'][ra:rb])
        c_off.append(block['run_off'][a:b] - ra + runs_before)
        c_runs.append(np.diff(block['run_off'][a:b + 1]))
        rids.append(block['rid'][a:b])
        offs.append(block['row_off'][lo + 1:hi + 1] - a + rows_before)
        runs_before, rows_before = runs_before + (rb - ra), rows_before + (b - a)
        as_int = as_int and block['score_is_int']
    cat = lambda parts, dt: np.concatenate(parts).astype(dt, copy=False) if parts else np.zeros(0, dt)
    arena = cat(arenas, np.uint32)
    return N.store_mat_member([cat(c, np.float64 if k in (2, 10, 11) else np.int64) for k, c in enumerate(cols)] +
                              [arena if len(arena) else np.zeros(1, np.uint32), cat(c_off, np.int64), cat(c_runs, np.int64), cat(rids, np.int64)],
                              cat(offs, np.int64), as_int)


def _emit_seq(take):
    from . import _native as N
    data, offs, before = [], [np.zeros(1, np.int64)], 0
    for (packed, pack_off), lo, hi in take:
        a, b = int(pack_off[lo]), int(pack_off[hi])
        data.append(packed[a:b])
        offs.append(pack_off[lo + 1:hi + 1] - a + before)
        before += b - a
    return N.store_seq_member(np.concatenate(data) if before else np.zeros(1, np.uint8), np.concatenate(offs))


def _slice_mat(block, lo, hi):
    """groups [lo, hi) of a _mat_block as a block of their own"""
    a, b = int(block['row_off'][lo]), int(block['row_off'][hi])
    ra, rb = int(block['run_off'][a]), int(block['run_off'][b])
    return dict(cols=[c[a:b] for c in block['cols']], arena=block['arena'][ra:rb], run_off=block['run_off'][a:b + 1] - ra, rid=block['rid'][a:b],
                row_off=block['row_off'][lo:hi + 1] - a, score_is_int=block['score_is_int'])


def _slice_seq(part, lo, hi):
    packed, pack_off = part
    a, b = int(pack_off[lo]), int(pack_off[hi])
    return packed[a:b], pack_off[lo:hi + 1] - a


def round_members(blocks, taxa, first, save_seq):
    """What the stores take from a ROUND of genomes whose first group has the store-wide id `first` (blocks: the genomes' StoreBlocks in
    job order, taxa: their taxon ids), made where the round was mapped - a worker process:
      n, first        groups of the round, id of its first
      table           int64[n, 7] gene-table rows, ids store-wide, taxon filled in
      c_src, c_val    the conflicts, ids store-wide
      mat, seq        per member store: head - column blocks [(block, groups)] of the ids in front of the round's first member boundary (they
                      complete the member that is open in front of the round) -, first_member + members - every member that lies inside the round,
                      FINISHED (payload, crc, size, method: pickle stream emitted, deflated) -, tail - the ids behind the last boundary"""
    n_each = [B.n for B in blocks]
    end = first + sum(n_each)
    b0 = min(end, -(-first // CHUNK) * CHUNK)
    b1 = max(b0, end // CHUNK * CHUNK)
    base = np.concatenate([[first], first + np.cumsum(n_each)]).tolist()

    def takes(parts, a, b):
        out = []
        for part, lo_id, n in zip(parts, base, n_each):
            lo, hi = max(a, lo_id), min(b, lo_id + n)
            if lo < hi:
                out.append((part, lo - lo_id, hi - lo_id))
        return out

    def store(parts, cut, emit, strategy):
        jobs = [_packers().submit(_pack_member, (lambda take: lambda: emit(take))(takes(parts, m, m + CHUNK)), strategy) for m in range(b0, b1, CHUNK)]
        return dict(head=[(cut(part, lo, hi), hi - lo) for part, lo, hi in takes(parts, first, b0)], first_member=b0 // CHUNK,
                    members=[j.result() for j in jobs], tail=[(cut(part, lo, hi), hi - lo) for part, lo, hi in takes(parts, b1, end)])
    rows, src, val = [], [], []
    for B, taxon, lo_id in zip(blocks, taxa, base):
        if B.n == 0:
            continue
        B.table[:, 1] = taxon
        B.table[:, 5] += lo_id
        rows.append(B.table)
        src.append(B.c_src + lo_id)
        val.append(B.c_val + 10 * lo_id)
    z = np.zeros(0, dtype=np.int64)
    return dict(n=end - first, first=first, table=np.vstack(rows) if rows else np.zeros([0, 7], dtype=np.int64), c_src=np.concatenate(src) if src else z, c_val=np.concatenate(val) if val else z,
                mat=store([B.mat for B in blocks], _slice_mat, _emit_mat, MAT_STRATEGY),
                seq=store([(B.packed, B.pack_off) for B in blocks], _slice_seq, _emit_seq, SEQ_STRATEGY) if save_seq else None)


class StoreBlock(object):
    """What the four stores take from ONE genome, with group ids still local to the genome (0 .. n-1): everything about a genome's groups
    that does not depend on the genomes in front of it.  Made where the groups are made - by a worker process when there are workers - so
    that the process that keeps the stores only adds the genome's first group id and cuts members:
      table             int64[n, 7]   the gene table's rows [gene, taxon (filled in by the keeper), score, identity, identity, LOCAL group id,
                                      fragments], x 1e4 where fractional, best score first in the order the reference's object sort gives
      mat               columns of the stored hit rows (_mat_block)
      packed, pack_off  the packed alleles
      c_src, c_val      the conflicts as the doubled, sorted list: group -> partner * 10 + class, both LOCAL"""
    __slots__ = ('n', 'contig', 'table', 'mat', 'packed', 'pack_off', 'c_src', 'c_val')

    def __init__(self, G):
        n = self.n = len(G)
        self.contig = int(G.contig[0]) if n else -1
        self.packed, self.pack_off = G.packed, G.pack_off
        self.mat = _mat_block(G) if n else None
        s4 = G.score * 10000
        order = np.argsort(-s4, kind='stable')
        if n > 1 and (np.diff(s4[order]) == 0).any():
            from ._native import argsort_object_order
            order = argsort_object_order(-s4)        # equal scores: the order among them is the one the reference's sort of its OBJECT column gives (numpy itself: 3 ms for 6 600 Python floats)
        i4 = (G.iden * 10000).astype(np.int64)
        rows = np.stack([G.gene, np.zeros(n, dtype=np.int64), s4.astype(np.int64), i4, i4, np.arange(n, dtype=np.int64),
                         np.diff(G.row_off).astype(np.uint8).astype(np.int64)], axis=1)
        self.table = rows[order]
        pairs = np.asarray(G.ovl, dtype=np.int64)
        src = np.concatenate([pairs[:, 0], pairs[:, 1]])
        val = np.concatenate([pairs[:, 1], pairs[:, 0]]) * 10 + np.concatenate([pairs[:, 2], pairs[:, 2]])
        order = np.argsort(src)                      # (the sort the reference applies to the doubled list: equal ids keep ITS order of ties)
        self.c_src, self.c_val = src[order], val[order]

    def __getstate__(self):
        return tuple(getattr(self, k) for k in self.__slots__)

    def __setstate__(self, state):
        for k, v in zip(self.__slots__, state):
            setattr(self, k, v)


def _stable_order_of_ids(ids):
    """np.argsort(ids, kind='stable') for non-negative integer ids: numpy sorts 16-bit keys by radix (stable, linear) - one pass below
    2^16, two below 2^32 (low half, then high half), against a merge sort of the 64-bit column that takes six times as long"""
    top = int(ids.max()) if len(ids) else 0
    if len(ids) == 0 or int(ids.min()) < 0 or top >= 1 << 32:
        return np.argsort(ids, kind='stable')
    order = np.argsort((ids & 0xFFFF).astype(np.uint16), kind='stable')
    if top >= 1 << 16:
        order = order[np.argsort((ids[order] >> 16).astype(np.uint16), kind='stable')]
    return order


class _StoreWriter(object):
    """what get_map_bsn keeps between genomes: the group counter and the unwritten parts of the four stores"""

    def __init__(self, conn, seq_conn, mat_conn, clf_conn, save_seq):
        self.conn = conn
        self.n_group, self.table, self.table_rows, self.t_table = 0, None, 0, 0.      # table: ONE growing int64 block the rows are copied into as they arrive
        self.conflicts = _ConflictBlocks(clf_conn)
        self.seqs = _MemberQueue(seq_conn, _emit_seq, SEQ_STRATEGY) if save_seq else None
        self.mats = _MemberQueue(mat_conn, _emit_mat, MAT_STRATEGY)

    def add(self, G, taxon):
        """the next genome's groups: a StoreBlock, or the GenomeGroups one is made from"""
        B = G if isinstance(G, StoreBlock) else StoreBlock(G)
        n, first = B.n, self.n_group
        if n == 0:
            return
        self.n_group += n
        if len(B.c_src):
            self.conflicts.add_sorted(B.c_src + first, B.c_val + 10 * first)
            self.conflicts.write(below=self.n_group)
        if self.seqs is not None:
            self.seqs.add((B.packed, B.pack_off), n)
        self.mats.add(B.mat, n)
        rows = B.table
        rows[:, 1] = taxon
        rows[:, 5] += first
        self._take_rows(rows)
        self.table_rows += n
        if self.table_rows >= TABLE_ROWS:
            self.write_table()

    def add_round(self, P):
        """the next round of genomes as round_members made it"""
        if P['first'] != self.n_group:
            raise RuntimeError('a round of genomes starts at group %d, the stores are at %d' % (P['first'], self.n_group))
        if P['n'] == 0:
            return
        self.n_group += P['n']
        if len(P['c_src']):
            self.conflicts.add_sorted(P['c_src'], P['c_val'])
            self.conflicts.write(below=self.n_group)
        for q, part in ((self.seqs, P['seq']), (self.mats, P['mat'])):
            if q is None:
                continue
            for block, n in part['head']:
                q.add(block, n)
            q.add_finished(part['first_member'], part['members'])
            for block, n in part['tail']:
                q.add(block, n)
        self._take_rows(P['table'])
        self.table_rows += P['n']
        if self.table_rows >= TABLE_ROWS:
            self.write_table()

    def _take_rows(self, rows):
        """rows of the gene table into the block that goes to the store at the end (a list of per-round blocks cost a copy of everything - 740 MB for 2 000
        genomes - when nobody else had anything left to do)"""
        rows = np.asarray(rows, dtype=np.int64)
        if rows.ndim != 2 or len(rows) == 0:
            return
        if self.table is None:
            self.table, self._filled = np.empty((max(1 << 16, 2 * len(rows)), rows.shape[1]), dtype=np.int64), 0
        need = self._filled + len(rows)
        if need > len(self.table):
            grown = np.empty((max(need, 2 * len(self.table)), self.table.shape[1]), dtype=np.int64)
            grown[:self._filled] = self.table[:self._filled]
            self.table = grown
        self.table[self._filled:need] = rows
        self._filled = need

    def write_table(self):
        if self.table is None or self._filled == 0:
            return
        t0 = time.perf_counter()
        tab = self.table[:self._filled]
        self.conn.update_table(tab, order=_stable_order_of_ids(tab[:, 0]))
        self.table, self.table_rows = None, 0
        self.t_table += time.perf_counter() - t0

    def close(self):
        if self.seqs is not None:           # (the last, partial members first: they are deflated and appended by the stores' threads while the gene table is made)
            self.seqs.close()
        self.mats.close()
        self.conflicts.write()
        self.write_table()


def get_map_bsn(prefix, clust, genomes, orthoGroup, old_prediction, conn, seq_conn, mat_conn, clf_conn, saveSeq, params, search=None, ctx=None,
                group=None, genomes_per_round=64, timing=None, workers=None):
    """genomes: {contig id: [taxon id, sequence]} -> fills the four MapBsn stores like PEPPAN.py:907-989:
      conn      gene id -> int rows [gene, taxon, score*1e4, ident*1e4, ident*1e4, group id, n fragments], best score first
      seq_conn  chunk no -> object array of packed alleles (only with saveSeq)
      mat_conn  chunk no -> object array of the hit rows of each group
      clf_conn  block no -> CSR [30001 offsets + 30001, partner*10 + class] of group-overlap conflicts
    `search(prefix, clust, jobs, params)` yields (blastab, overlap) per genome in job order (default: batched GPU search).
    Under torch.distributed (one process per GPU) the genomes are sharded over the ranks in blocks of `genomes_per_round`; rank 0
    writes the stores (the other ranks pass None for them), whose contents do not depend on the number of ranks.
    No Python object is made for a stored hit row: the groups of a genome stay columns (GenomeGroups) and the members of the .mat /
    .seq stores are emitted from them as .npy pickle streams by host C++ on the stores' writer threads.
    `workers`: a number of worker processes or an open mapworkers.MapWorkers - the reference's pool of forked workers (PEPPAN.py:922):
    rounds of genomes are searched and grouped by the workers, each with a HIP context of its own on this process's device; they also make
    the .mat / .seq members that lie inside their rounds, and this process only appends to the stores.  A mapped genome costs ten times
    more host than GPU time: eight workers map five times as many genomes per second on one GPU.  Default (None): min(8, params['n_thread'])
    workers - the size of the reference's pool - from 64 genomes on when the caller's params carry `n_thread`, otherwise everything in
    this process (0 / 1 say so explicitly)."""
    if len(genomes) == 0:
        raise ValueError('get_map_bsn: no genome to map against')
    t_enter = time.perf_counter()
    taxa = {}
    for g, s in genomes.items():
        taxa.setdefault(s[0], []).append([g, s[1]])
    jobs = [(id, taxon, seq) for id, (taxon, seq) in enumerate(taxa.items())]
    ortho = OrthoRelation(orthoGroup)
    old_is_mine = isinstance(old_prediction, str)
    if old_is_mine:
        old_prediction = MapBsn(old_prediction)            # opened once, read by every genome (the reference opens it per genome, PEPPAN.py:870)
    per_round = max(1, int(genomes_per_round))
    searcher = search or (lambda *a: _gpu_search(*a, genomes_per_batch=per_round))
    pool, own_pool = None, False
    own_ctx = None
    if workers is None and search is None and len(jobs) >= 64:
        workers = min(8, int(params.get('n_thread', 0) or 0), effective_cpus() // 2)     # the reference's pool has n_thread workers (PEPPAN.py:1841); a GPU feeds about eight, a worker wants two CPUs
    if workers is not None and not (isinstance(workers, int) and workers <= 1):
        from .mapworkers import MapWorkers
        pool, own_pool = (workers, False) if isinstance(workers, MapWorkers) else (MapWorkers(int(workers), device=getattr(ctx, 'device', None)), True)     # (the workers' contexts: on the caller's device)
        if _dist_world(group)[1] == 1:     # (under torch.distributed `per_round` is the dealing unit of the RANKS - block k*world + rank, the number of gathers -
            #                                 and must be the same on every rank whatever CPUs each was granted: the caller's value stays)
            # four rounds per worker or more - the last ones even the load out - and never more than POOL_ROUND genomes: a worker searches its next round while it
            # groups the one in front, and rounds of 8 keep that pipeline and the order of the rounds' first ids tight (2 000 genomes: 454 genomes/s against 418 with
            # rounds of 16 and 354 with 32; 4 costs the searches' fixed launches more than it evens out)
            # - and of at most POOL_ROUND_NT nucleotides: 50 000 exemplars on genomes of 7.7 Mb map 109 genomes/s in rounds of 4 against 94 in rounds of 8 (and of 2)
            mean_nt = sum(len(sq) for job in jobs for _, sq in job[2]) / float(len(jobs))
            short = max(1, min(POOL_ROUND, int(POOL_ROUND_NT // max(1., mean_nt))))
            per_round = max(min(4, per_round, short), min(per_round, short, -(-len(jobs) // (4 * pool.n))))
        try:
            from . import _native
            pool.setup(prefix, clust, orthoGroup, old_prediction, params, search=search, per_batch=per_round,
                       ctx_class=None if ctx is None or isinstance(ctx, _native.Context) else type(ctx), save_seq=saveSeq,
                       form='members' if _dist_world(group)[1] == 1 else 'stores')     # (under torch.distributed a rank does not know its rounds' first ids)
        except BaseException:
            if own_pool:
                pool.close()
            if old_is_mine:
                old_prediction.close()
            raise
    if pool is None and search is None and ctx is None and _dist_world(group)[1] == 1 and len(jobs) > 1:
        # ONE process, the product's own search: the searches run on a thread of their own (_ahead), in batches of at most ONE_PROCESS_BATCH genomes so that
        # the second batch is being searched while the first gets its groups - and K12, inside build_groups, works in a context of its own (the shared one of
        # uberBlast.get_context belongs to the search thread)
        from . import uberBlast
        ctx = own_ctx = uberBlast.get_side_context('k12')              # (kept by the process like the search contexts: making and closing one costs 30 ms a call)
        mean_nt = sum(len(sq) for job in jobs for _, sq in job[2]) / float(len(jobs))
        batch = max(1, min(per_round, ONE_PROCESS_BATCH, int(POOL_ROUND_NT // max(1., mean_nt))))          # (large genomes: fewer per search, as in a pool's rounds)
        searcher = lambda *a: _gpu_search(*a, genomes_per_batch=batch)
        searcher.runs_ahead = True
    stores = _StoreWriter(conn, seq_conn, mat_conn, clf_conn, saveSeq)
    clock = time.perf_counter
    spent = dict(stores=0.)
    # The stores are fed by a thread of their own, genome by genome in job order: the caller's thread goes straight back to the next
    # genome's search / filters / K12 and never waits for an archive (handing a genome to the stores took 16 ms of a 44 ms genome while
    # it shared a thread with them: queues filling up behind the deflating writers).  At most two rounds of genomes wait in the queue.
    inbox, failure = queue.Queue(maxsize=2 * per_round), []

    def keeper():
        while True:
            item = inbox.get()
            if item is None:
                spent['keeper_cpu'] = time.thread_time()
                return
            if failure:
                continue                            # (keep draining so that the producer never blocks on a full queue)
            try:
                id, G = item
                t0 = clock()
                if isinstance(G, dict):             # a round of genomes whose members the workers made: id = the round's jobs
                    stores.add_round(G)
                    for job in id:
                        logger('Merged {0}.{1}'.format(prefix, job[0]))
                else:
                    B = G if isinstance(G, StoreBlock) else StoreBlock(G)
                    if B.n:
                        stores.add(B, genomes.get(B.contig, [-1])[0])
                    logger('Merged {0}.{1}'.format(prefix, id))
                spent['stores'] += clock() - t0
            except BaseException as e:
                failure.append(e)

    worker = threading.Thread(target=keeper, daemon=True)
    worker.start()
    t_start, main_cpu0 = clock(), time.thread_time()
    interval = sys.getswitchinterval()
    try:
        if own_ctx is not None:
            sys.setswitchinterval(1e-3)             # (one process: the search thread and this one hand the interpreter to each other; restored below whatever happens)
        for job, G in _all_groups(prefix, clust, jobs, ortho, old_prediction, params, searcher, ctx, group, per_round, pool):
            if failure:
                break
            inbox.put((job, G) if isinstance(G, dict) else (job[0], G))
    finally:
        t_groups = clock()
        inbox.put(None)
        worker.join()
        if own_pool:
            pool.close()
        if old_is_mine:
            old_prediction.close()
        sys.setswitchinterval(interval)
    if failure:
        raise failure[0]
    t0 = clock()
    stores.close()
    if timing is not None:      # seconds: search + filters + build_groups on the caller's thread; then, on the stores' thread and overlapped with it,
        #                         handing groups to the stores and the gene table updates; what was left to wait for at the end
        timing.update(setup=t_start - t_enter, close=clock() - t0, groups=t_groups - t_start, stores=spent['stores'], keeper_thread_cpu=spent.get('keeper_cpu', 0.), main_thread_cpu=time.thread_time() - main_cpu0, gene_table=stores.t_table, drain=clock() - t_groups)
