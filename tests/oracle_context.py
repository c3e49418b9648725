"""Test helper: an object with the interface of peppan_amd._native.Context whose every result comes from the CPU
oracle (oracle/).  Lets the GPU tests run the product's host code twice - once over the HIP library, once over
the oracle - and compare whole tables."""
import numpy as np
from oracle import oracle as O
from peppan_amd import _native as N


class OracleContext(object):
    @staticmethod
    def _texts(seqs):
        if isinstance(seqs, tuple) and len(seqs) == 2:                       # (bytes, offsets), the packed form Context.set_*_nt also takes
            buf, off = np.asarray(seqs[0], dtype=np.uint8).tobytes().decode('ascii'), np.asarray(seqs[1], dtype=np.int64)
            return [buf[off[i]:off[i + 1]] for i in range(len(off) - 1)]
        return [s.decode() if isinstance(s, bytes) else s for s in seqs]

    def set_query_nt(self, seqs, gtable=11):
        self.q_nt = self._texts(seqs)
        self.q_table = gtable
        self._direct = False
        self.q_aa_given = False

    def set_ref_nt(self, seqs, frames=6, gtable=11):
        self.r_nt = self._texts(seqs)
        self.r_frames, self.r_table = frames, gtable

    @staticmethod
    def _unpack(seqs):
        if isinstance(seqs, tuple) and len(seqs) == 2:                       # (codes, offsets), as Context._pack passes through
            codes, off = np.asarray(seqs[0], dtype=np.uint8), np.asarray(seqs[1], dtype=np.int64)
            return [codes[off[i]:off[i + 1]] for i in range(len(off) - 1)]
        return [np.asarray(s, dtype=np.uint8) for s in seqs]

    def set_query_aa(self, seqs):
        self.q_aa = self._unpack(seqs)
        self._direct = True
        self.q_aa_given = True

    def set_ref_aa(self, seqs):
        self.t_aa = self._unpack(seqs)
        self._direct = True
        self._raw_targets = True

    def alleles(self, contigs, rows, cigar, grp_off, grp_qlen, gtable=11):
        return O.alleles([c.encode('ascii') if isinstance(c, str) else bytes(c) for c in contigs], rows, cigar, grp_off, grp_qlen, gtable)

    def sha1(self, seqs):
        return O.sha1_digests(seqs)

    def dedup(self, lengths, digests):
        return O.dedup(lengths, digests)

    def set_target_groups(self, groups):
        assert groups is None or len(groups) == 0, 'the oracle searches one reference set at a time'

    def use_nt_as_residues(self, strands=2):
        """pep_use_nt_as_residues: base codes of the nucleotide sets; reference = all forward strands, then all reverse complements"""
        self.q_aa = [O.nt_codes(s.upper()) for s in self.q_nt]
        fwd = [O.nt_codes(s.upper()) for s in self.r_nt]
        rc = [np.where(c < 4, 3 - c, 4).astype(np.uint8)[::-1] for c in fwd]
        self.t_aa = fwd + (rc if strands == 2 else [])
        n = len(fwd)
        self._qm = np.array([(i, 1, len(c), len(c)) for i, c in enumerate(self.q_aa)], dtype=N.QUERY_META_DTYPE).reshape(-1)
        self._tm = np.array([(i % n, 1 if i < n else 4, 0, len(c)) for i, c in enumerate(self.t_aa)], dtype=N.TARGET_META_DTYPE).reshape(-1)
        self._direct = True
        self._raw_targets = False

    def translate(self, force=False):
        if hasattr(self, 'q_nt') and hasattr(self, 'r_nt') and getattr(self, 'q_aa_given', False) is False:
            self._direct = False            # (K1 again: a nucleotide search may have left base codes in the packed sets)
        self._translate_now()

    def _translate_now(self):
        if getattr(self, '_direct', False):
            return
        qm, self.q_aa = [], []
        for i, s in enumerate(self.q_nt):
            f, p = O.query_frame(s.upper(), self.q_table)
            qm.append((i, f, len(p), len(s)))
            self.q_aa.append(O.aa_codes(p))
        tm, self.t_aa = [], []
        for i, s in enumerate(self.r_nt):
            fl = range(1, 7) if self.r_frames == 6 else range(1, 4)
            for f, aa in zip(fl, O.translate_frames(s.upper(), fl, self.r_table)):
                for off, c in O.ref_chunks(aa):
                    tm.append((i, f, off, len(c)))
                    self.t_aa.append(O.aa_codes(c))
        self._qm = np.array(qm, dtype=N.QUERY_META_DTYPE).reshape(-1)
        self._tm = np.array(tm, dtype=N.TARGET_META_DTYPE).reshape(-1)
        self._raw_targets = False

    def query_meta(self):
        return self._qm

    def target_meta(self):
        return self._tm

    def search(self, params=None, copy=True):
        self._translate_now()
        p = O.params_from(params)
        ms = np.array([O.min_score(len(s), params.dbsize, params.max_evalue, params.ka_lambda, params.ka_k) for s in self.q_aa], dtype=np.int32)
        subjects = None
        if params.hsp_mode == 2:            # the reference sequence every target is a strand / frame / chunk of (the library takes it from K1's table)
            tm = getattr(self, '_tm', None)
            subjects = tm['seq'] if tm is not None and len(tm) == len(self.t_aa) and not getattr(self, '_raw_targets', False) else np.arange(len(self.t_aa))
        h, c, st = O.search(self.q_aa, self.t_aa, p, min_scores=ms, subjects=subjects)
        out = np.zeros(len(h), dtype=N.HIT_DTYPE)
        for f in out.dtype.names:
            out[f] = h[f]
        return out, c, st

    def rescore_nt(self, h, arena):
        out = np.zeros((len(h), 5), dtype=np.int64)
        enc_q, enc_r = {}, {}                          # a genome-sized reference is encoded once, not once per hit
        for k in range(len(h)):
            qi, ri = int(h['q'][k]), int(h['r'][k])
            if qi not in enc_q:
                enc_q[qi] = O.nt_encode_rescore(self.q_nt[qi].upper())
            if ri not in enc_r:
                enc_r[ri] = O.nt_encode_rescore(self.r_nt[ri].upper())
            out[k] = O.rescore_counts(enc_q[qi], enc_r[ri], int(h['qs'][k]), int(h['rs'][k]), int(h['re'][k]),
                                      arena[int(h['cigar_off'][k]):int(h['cigar_off'][k]) + int(h['cigar_runs'][k])])
        return out

    def pair_support(self, rows, cigar, grp_off, grp_qlen, grp_rlen, limits):
        lim = dict(match_len=list(limits.match_len), match_prop=list(limits.match_prop), identity_x1e4=limits.identity_x1e4, any_frame=bool(limits.any_frame))
        out = np.full(len(grp_qlen), N.SUPPORT_NONE, dtype=np.int32)
        for g in range(len(grp_qlen)):
            rr = []
            for k in range(int(grp_off[g]), int(grp_off[g + 1])):
                o, c = int(rows['cigar_off'][k]), int(rows['cigar_runs'][k])
                rr.append((int(rows['q_start'][k]), int(rows['r_start'][k]), float(rows['identity'][k]), [(int(x) >> 2, int(x) & 3) for x in cigar[o:o + c]]))
            v = O.pair_support(rr, int(grp_qlen[g]), int(grp_rlen[g]), lim) if rr else None
            if v is not None:
                out[g] = v
        return out

    def components(self, n, a, b):
        return O.components(n, a, b)

    def overlaps(self, contig, start, end, row_id, ovl_l, ovl_p):
        return O.overlaps_sweep(contig, start, end, row_id, ovl_l, ovl_p)
