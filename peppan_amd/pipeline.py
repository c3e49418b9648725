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

from .clust import getClust
from .configure import logger, readFasta
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


def identity_schedule(target):
    """the identity levels of the iterative clustering: 1.00, 0.99, ... down to the target (11 levels for 0.90)"""
    return [float(x) for x in np.round(np.arange(1., target - 0.005, -0.01), 5)]


def _tab_pairs_after_first_line(tab):
    """(gene, exemplar) rows of a clust.tab file WITHOUT its first line, columns typed the way a header-inferring CSV reader
    types them (all-integer column -> ints).  The reference reads the file with pandas' default header handling
    (PEPPAN.py:1786), so the first gene of every table silently never reaches clust.npy; downstream results depend on it."""
    with open(tab) as fin:
        rows = [line.rstrip('\n').split('\t')[:2] for line in fin][1:]
    cols = [[r[0] for r in rows], [r[1] for r in rows]]
    for c in cols:
        if c and all(re.fullmatch(r'[+-]?\d+', x) for x in c):
            c[:] = [int(x) for x in c]
    return list(zip(*cols))


def iterClust(prefix, genes, geneGroup, params):
    """clustering at falling identity levels, each level on the exemplars of the one before; appends [exemplar, member,
    identity label] edges to geneGroup, saves them as <prefix>.clust.npy and returns the last exemplar file"""
    coverage = np.round(params['coverage'], 2)
    current = genes
    for level in identity_schedule(params['identity']):
        params.update(identity=level, coverage=coverage)
        current, tab = getClust(prefix, current, params)
        exemplars = readFasta(current, headOnly=True)
        logger('Iterative clustering. {0} exemplars left with identity = {1}'.format(len(exemplars), level))
        label = int(min(1., level + 0.005) * 10000)
        for gene, other in _tab_pairs_after_first_line(tab):
            if gene != other:
                geneGroup.append([gene, other, label] if str(gene) in exemplars else [other, gene, label])
    np.save('{0}.clust.npy'.format(prefix), np.array(geneGroup, dtype=int))
    return current


# ---- the decision pass over the all-vs-all table (PEPPAN.py:194-294; behaviour spec: SURVEY.md appendix A.4) -----------------
ORDINARY, CONFLICT, ABSORB_QUERY, ABSORB_REF = 0, 1, 2, 3
_CIGAR_RUN = re.compile(r'(\d+)([A-Z])')


class _SupportLimits(object):
    """the thresholds of the ortholog-support test, derived once from the parameter dict"""

    def __init__(self, params):
        self.lens = (params['match_len'], params['match_len1'], params['match_len2'])
        self.props = (params['match_prop'], params['match_prop1'], params['match_prop2'])
        self.identity = params['match_identity'] * 10000
        self.any_frame = 'f' in params['incompleteCDS']

    def enough_to_decide(self, n_nt, q_len):
        return n_nt >= min(self.lens) and n_nt >= min(self.props) * q_len

    def full_support(self, n_nt, shorter):
        return n_nt >= min(max(l, p * shorter) for l, p in zip(self.lens, self.props))


def _pair_support(hits, lim):
    """Ortholog support of one (query, reference) pair from its forward hits in table order.
    A coverage map over the query's nucleotide positions records, for every position inside an in-frame M run (counted from the
    first position of the run that starts a codon of the query), the identity of the LAST hit that covered it; positions are remembered in the
    order they were first covered, because the mean is taken over that sequence.  After every M run: once enough positions
    are covered the mean identity is tested - passing ends the walk with the mean (full support) or 0 (partial support);
    failing lets later runs and hits add positions.  -> None (no decision) | 0 | int(mean identity * 1e4)"""
    q_len, r_len = int(hits[0][12]), int(hits[0][13])
    if 20 * min(q_len, r_len) <= max(q_len, r_len):
        return None
    parsed = [(int(h[6]), int(h[8]), h[2], [(int(n), op) for n, op in _CIGAR_RUN.findall(h[14])]) for h in hits]
    size = max(q0 + sum(n for n, op in runs if op in 'MI') for q0, _, _, runs in parsed) + 1
    covered = np.zeros(size, dtype=bool)
    ident_at = np.zeros(size, dtype=np.float64)
    first_cover_order, n_cov = [], 0
    for qpos, rpos, ident, runs in parsed:
        for n, op in runs:
            if op == 'I':
                qpos += n
            elif op != 'M':
                rpos += n
            else:
                lo, hi = qpos + (1 - qpos) % 3, qpos + n
                if lo < hi and (lim.any_frame or qpos % 3 == rpos % 3):
                    fresh = np.flatnonzero(~covered[lo:hi]) + lo
                    first_cover_order.append(fresh)
                    n_cov += fresh.size
                    covered[lo:hi] = True
                    ident_at[lo:hi] = ident
                qpos += n
                rpos += n
                if lim.enough_to_decide(3 * n_cov, q_len):
                    mean = int(np.mean(ident_at[np.concatenate(first_cover_order)]) * 10000)
                    if mean >= lim.identity:
                        return mean if lim.full_support(3 * n_cov, min(q_len, r_len)) else 0
    return None


def _classify_rows(table, priorities, near_identity, cover):
    """order-independent part of the decision, for all rows at once -> (action code per row, forward flag per row)"""
    num = lambda c: table[:, c].astype(np.float64)
    q, r = table[:, 0].astype(np.int64), table[:, 1].astype(np.int64)
    iden, qs, qe, ss, se, ql, sl = num(2), num(6), num(7), num(8), num(9), num(12), num(13)
    rank_q = np.array([priorities[g][0] for g in q.tolist()])
    rank_r = np.array([priorities[g][0] for g in r.tolist()])
    q_span, r_span = qe - qs + 1, np.abs(se - ss) + 1
    near = (q != r) & (iden >= near_identity)
    same_head, same_tail = (qs % 3 == ss % 3), ((ql - qe) % 3 == (sl - se) % 3)
    off_frame = (ss > se) | (~same_head & same_tail)                              # reverse strand, or shifted at the head only
    in_frame = ~off_frame & (ss < se) & same_head & same_tail
    root = np.sqrt(cover)
    action = np.full(len(table), ORDINARY, dtype=np.int8)
    action[near & off_frame & ((q_span >= cover * ql) | (r_span >= cover * sl))] = CONFLICT
    action[near & in_frame & (ql <= sl) & (q_span >= root * sl) & (rank_q >= rank_r)] = ABSORB_QUERY
    action[near & in_frame & (ql > sl) & (r_span >= root * ql) & (rank_q <= rank_r)] = ABSORB_REF
    return q.tolist(), r.tolist(), action.tolist(), (ss < se).tolist(), (iden * 10000.).astype(np.int64).tolist()


def _self_search(clust, params, pool):
    tools = '--blastn' if params['noDiamond'] else '--blastn --diamond -s 1'
    argv = ['-r', clust, '-q', clust] + tools.split() + ['--min_id', str(params['match_identity'] - 0.05), '--min_cov', str(params['match_frag_len']),
                                                        '-t', str(params['n_thread']), '--min_ratio', str(params['match_frag_prop']), '-e', '3,3', '-p',
                                                        '--gtable', str(params['gtable'])]
    return uberBlast(argv, pool)


def get_similar_pairs(clust, priorities, params, pool=None):
    """All-vs-all search of the exemplars, then ONE ordered pass over its table (rows sorted by query, reference, score with
    the names compared as strings) deciding which exemplars are absorbed by a near-identical in-frame partner, which pairs
    conflict (-2) and which are ortholog-like (mean identity * 1e4).  The row-local tests are evaluated for all rows at once
    (_classify_rows); only the state that makes the pass order-dependent is walked row by row:
      alive[g]   0 once g was absorbed or found repetitive; rows touching a dead gene are ignored from then on
      pending    the forward rows of the current (query, reference) pair; the pair is settled when the NEXT surviving row
                 belongs to another pair (or at the end) - not earlier, rows in between still see the old state
    Side effects as in the reference: the exemplar FASTA loses the dead genes, absorbed pairs are added to clust.npy."""
    table = _self_search(clust, params, pool)
    if len(table):
        table.T[:2] = table.T[:2].astype(int)
    q, r, action, forward, iden4 = _classify_rows(table, priorities, params['clust_identity'], params['clust_match_prop']) if len(table) else ([], [], [], [], [])
    lim = _SupportLimits(params)
    alive, verdict, absorbed, pending = {}, {}, [], []

    def settle(rows):
        a, b = q[rows[0]], r[rows[0]]
        if len(rows) >= 50:
            alive[b] = 0                                   # fifty or more hits between two genes: the reference gene is a repeat
        elif a != b and (min(a, b), max(a, b)) not in verdict:
            value = _pair_support([table[k] for k in rows], lim)
            if value is not None:
                verdict[(min(a, b), max(a, b))] = value

    for k in range(len(table)):
        a, b = q[k], r[k]
        if alive.setdefault(a, 1) == 0 or alive.get(b, 1) == 0:
            continue
        if action[k] == CONFLICT:
            verdict[(min(a, b), max(a, b))] = -2
        elif action[k] == ABSORB_QUERY:
            absorbed.append([b, a, iden4[k]])
            alive[a] = 0
        elif action[k] == ABSORB_REF:
            absorbed.append([a, b, iden4[k]])
            alive[b] = 0
        elif forward[k]:
            if pending and (q[pending[0]], r[pending[0]]) != (a, b):
                settle(pending)
                pending = []
            pending.append(k)
    if pending:
        settle(pending)

    _drop_dead_exemplars(params['clust'], alive)
    if absorbed:
        npy = params['clust'].rsplit('.', 1)[0] + '.npy'
        edges = np.vstack([np.load(npy, allow_pickle=True), absorbed])
        np.save(npy, edges[np.argsort(-edges.T[2])])
    return np.array([[a, b, v] for (a, b), v in verdict.items() if v != 0], dtype=int)


def _drop_dead_exemplars(fasta, alive):
    """rewrite the exemplar FASTA in place, keeping the records of genes that appeared as a query and are still alive"""
    from .clust import read_blocks
    keep = [blk.text for blk in read_blocks(fasta) if alive.get(int(blk.name), 0) > 0]
    with open(fasta, 'w') as fout:
        fout.writelines(keep)


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
