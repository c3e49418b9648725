"""The callers either side of the search in PEPPAN.py, with the same inputs, outputs and on-disk artefacts
(SURVEY.md section 8a rows a14-a17), so that the GPU path plugs into an unmodified pipeline run:

    writeGenes(fname, genes, priority)            PEPPAN.py:1023-1039   -> <prefix>.genes + exact-duplicate groups
    iterClust(prefix, genes, geneGroup, params)   PEPPAN.py:1777-1792   -> <prefix>.clust.exemplar/.tab/.npy
    get_similar_pairs(clust, priorities, params)  PEPPAN.py:194-294     -> int[n,3] ortholog / conflict pairs,
                                                                          rewrites the exemplar FASTA, extends clust.npy
    get_gene_group(cluFile, bsnFile)              PEPPAN.py:1590-1609   -> {root: [members in merge order]}
    gene_group_labels(cluFile, bsnFile, n)        same partition as get_gene_group, as labels from the GPU (K10)
"""
import re
from operator import itemgetter

import numpy as np
import pandas as pd

from .clust import getClust
from .configure import logger, readFasta, uopen
from .uberBlast import uberBlast, get_context


def gene_hashes(seqs, ctx=None):
    """the integer PEPPAN stores per gene and breaks priority ties with: int(hashlib.sha1(seq).hexdigest(), 16)
    (PEPPAN.py:62, 1019), computed for the whole list on the GPU (K13 `pep_sha1`)"""
    ctx = ctx or get_context()
    return [int.from_bytes(d.tobytes(), 'big') for d in ctx.sha1(seqs)]


def writeGenes(fname, genes, priority, ctx=None):
    """genes in priority order; a gene whose (length, sha1) equals an already written one of the SAME length run is
    reported as its duplicate.  The reference rebuilds its seen-table whenever a length not currently in it shows up
    (PEPPAN.py:1032-1033), so duplicates are only found while one length is 'open'.  The collapse itself runs on the GPU
    (K13 `pep_dedup`: smallest priority index per (length run, digest))."""
    ctx = ctx or get_context()
    order = [n for n, _ in sorted(priority.items(), key=itemgetter(1)) if n in genes and len(genes[n][6])]
    lengths = np.array([len(genes[n][6]) for n in order], dtype=np.uint32)
    digests = np.frombuffer(b''.join(int(genes[n][5]).to_bytes(20, 'big') for n in order), dtype=np.uint8).reshape(-1, 20)
    rep = ctx.dedup(lengths, digests).tolist()
    groups = []
    with open(fname, 'w') as fout:
        for i, n in enumerate(order):
            if rep[i] == i:
                fout.write('>{0}\n{1}\n'.format(n, genes[n][6]))
            else:
                groups.append([order[rep[i]], n, 10000])
    return fname, groups


def iterClust(prefix, genes, geneGroup, params):
    """11 clustering steps from identity 1.00 down to the target, each on the previous step's exemplars"""
    target = params['identity']
    g = genes
    for iden in np.round(np.arange(1., target - 0.005, -0.01), 5):
        params.update({'identity': iden, 'coverage': np.round(params['coverage'], 2)})
        label = int(min(1., iden + 0.005) * 10000)
        g, tab = getClust(prefix, g, params)
        exemplars = readFasta(g, headOnly=True)
        # NB: no header=None - pandas takes the first line of clust.tab as column names, exactly like PEPPAN.py:1786
        pairs = pd.read_csv(tab, sep='\t').values
        logger('Iterative clustering. {0} exemplars left with identity = {1}'.format(len(exemplars), iden))
        for g1, g2 in pairs[pairs.T[0] != pairs.T[1]]:
            r, q = (g1, g2) if str(g1) in exemplars else (g2, g1)
            geneGroup.append([r, q, label])
    np.save('{0}.clust.npy'.format(prefix), np.array(geneGroup, dtype=int))
    return g


_CIGAR_RUN = re.compile(r'(\d+)([A-Z])')


def _ortholog_value(rows, ortho_pairs, params):
    """decision for one (query, reference) pair from all of its forward hits (PEPPAN.py:195-224)"""
    key = tuple(sorted([rows[0][0], rows[0][1]]))
    if key in ortho_pairs:
        return
    ql, sl = int(rows[0][12]), int(rows[0][13])
    if min(ql, sl) * 20 <= max(ql, sl):
        return
    need_len = min(params['match_len2'], params['match_len'], params['match_len1'])
    need_prop = min(params['match_prop'], params['match_prop1'], params['match_prop2']) * ql
    any_frame = 'f' in params['incompleteCDS']
    matched = {}
    for part in rows:
        qpos, _, spos, _ = [int(x) for x in part[6:10]]
        for n, op in _CIGAR_RUN.findall(part[14]):
            n = int(n)
            if op == 'M':
                fq, fs = qpos % 3, spos % 3
                if fq == fs or any_frame:
                    matched.update({qpos + x: part[2] for x in range((3 - (fq - 1)) % 3, n)})
                qpos += n
                spos += n
                if len(matched) * 3 >= need_len and len(matched) * 3 >= need_prop:
                    ave = int(np.mean(list(matched.values())) * 10000)
                    if ave >= params['match_identity'] * 10000:
                        short = min(sl, ql)
                        full = min(max(params['match_len'], params['match_prop'] * short), max(params['match_len1'], params['match_prop1'] * short),
                                   max(params['match_len2'], params['match_prop2'] * short))
                        ortho_pairs[key] = ave if len(matched) * 3 >= full else 0
                        return
            elif op == 'I':
                qpos += n
            else:
                spos += n


def get_similar_pairs(clust, priorities, params, pool=None):
    """all-vs-all search of the exemplars and the single ordered pass over its table that decides which exemplars
    are absorbed (near-identical, in frame), which pairs conflict (-2) and which are ortholog-like (identity*1e4)"""
    flags = '--blastn' if params['noDiamond'] else '--blastn --diamond -s 1'
    self_bsn = uberBlast('-r {0} -q {0} {6} --min_id {1} --min_cov {2} -t {3} --min_ratio {4} -e 3,3 -p --gtable {5}'.format(
        clust, params['match_identity'] - 0.05, params['match_frag_len'], params['n_thread'], params['match_frag_prop'], params['gtable'], flags).split(), pool)
    self_bsn.T[:2] = self_bsn.T[:2].astype(int)
    presence, ortho_pairs, absorbed, buf = {}, {}, [], []
    ci, cmp_ = params['clust_identity'], params['clust_match_prop']

    def flush():
        if len(buf) >= 50:
            presence[buf[0][1]] = 0                    # >= 50 hits for one pair: the reference gene is repetitive
        elif buf and buf[0][0] != buf[0][1]:
            _ortholog_value(buf, ortho_pairs, params)

    for part in self_bsn:
        q, r = part[0], part[1]
        if q not in presence:
            presence[q] = 1
        elif presence[q] == 0:
            continue
        iden, qs, qe, ss, se, ql, sl = float(part[2]), float(part[6]), float(part[7]), float(part[8]), float(part[9]), float(part[12]), float(part[13])
        if presence.get(r, 1) == 0:
            continue
        qa, sa = qe - qs + 1, abs(se - ss) + 1
        if q != r and iden >= ci:
            same_tail = (ql - qe) % 3 == (sl - se) % 3
            if ss > se or (qs % 3 != ss % 3 and same_tail):
                if qa >= cmp_ * ql or sa >= cmp_ * sl:
                    ortho_pairs[tuple(sorted([q, r]))] = -2
                    continue
            elif ss < se and qs % 3 == ss % 3 and same_tail:
                if ql <= sl:
                    if qa >= np.sqrt(cmp_) * sl and priorities[q][0] >= priorities[r][0]:
                        absorbed.append([int(r), int(q), int(iden * 10000.)])
                        presence[q] = 0
                        continue
                elif sa >= np.sqrt(cmp_) * ql and priorities[q][0] <= priorities[r][0]:
                    absorbed.append([int(q), int(r), int(iden * 10000.)])
                    presence[r] = 0
                    continue
        if ss >= se:
            continue
        if buf and (buf[0][0] != q or buf[0][1] != r):
            flush()
            buf = []
        buf.append(part)
    if buf:
        flush()

    kept = []
    with uopen(params['clust'], 'r') as fin:
        write = False
        for line in fin:
            if line.startswith('>'):
                write = presence.get(int(line[1:].strip().split()[0]), 0) > 0
            if write:
                kept.append(line)
    with open(params['clust'], 'w') as fout:
        fout.writelines(kept)
    if absorbed:
        npy = params['clust'].rsplit('.', 1)[0] + '.npy'
        clu = np.vstack([np.load(npy, allow_pickle=True), absorbed])
        np.save(npy, clu[np.argsort(-clu.T[2])])
    return np.array([[k[0], k[1], v] for k, v in ortho_pairs.items() if v != 0], dtype=int)


def _edges(cluFile, bsnFile):
    clu = np.load(cluFile.rsplit('.', 1)[0] + '.npy', allow_pickle=True)
    bsn = np.load(bsnFile, allow_pickle=True)
    return clu, bsn[bsn.T[2] > 0]


def get_gene_group(cluFile, bsnFile):
    """single-linkage groups: the q side's whole group is appended to the r side's group and re-tagged with its root
    (merge order is kept - downstream code depends on it)"""
    groups, tag = {}, {}
    for matrix in _edges(cluFile, bsnFile):
        for r, q, _ in matrix:
            q_root, r_root = tag.get(q, q), tag.get(r, r)
            if q_root == r_root:
                continue
            moved = groups.pop(q_root, [q_root])
            for x in moved:
                tag[x] = r_root
            groups[r_root] = groups.get(r_root, [r_root]) + moved
    return groups


def gene_group_labels(cluFile, bsnFile, n_genes=None, device=None):
    """the same partition as get_gene_group, computed on the GPU (K10): label[g] = smallest gene id of g's group"""
    clu, bsn = _edges(cluFile, bsnFile)
    e = np.vstack([clu[:, :2], bsn[:, :2]]).astype(np.int64) if len(clu) + len(bsn) else np.zeros((0, 2), np.int64)
    n = int(n_genes if n_genes is not None else (e.max() + 1 if len(e) else 0))
    return get_context(device).components(n, e[:, 0], e[:, 1])
