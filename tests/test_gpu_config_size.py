"""Config-size runs on the MI355X: the stages of BASELINE.json's configs that round 1 only touched at toy sizes.
  (a) configs[2]/[3] mapping stage: 10 000 exemplar genes x 64 synthetic genomes through get_map_bsn (PEPPAN.py:907-989), batched GPU search;
      properties on all genomes, whole-table equality with the oracle-driven host code on a sample of them
  (b) configs[2] front end: writeGenes + iterClust (PEPPAN.py:1023-1039, 1777-1792) on 1 M gene instances, time and memory asserted
      (tools/front_end_scale.py runs the same at 5 M)
  (c) configs[4] search stage: 50 000 x 50 000 genes all-vs-all, bit-exact against the OpenMP oracle
"""
import contextlib
import io
import os
import resource
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    from peppan_amd import _native as N
    with N.Context(0) as c:
        yield c


def _cmp_tables(a, b):
    assert a.shape == b.shape, (a.shape, b.shape)
    assert a.tolist() == b.tolist()


@pytest.mark.parametrize('n_genomes', [64, 500])
def test_map_bsn_10k_exemplars_x_genomes(tmp_path, monkeypatch, n_genomes):
    """BASELINE configs[2] mapping stage: 10 000 exemplars against 64 genomes, and at the configuration's FULL size - 500 genomes, 1.07 Gnt -
    through get_map_bsn; every planted allele found, stores consistent, sampled genomes equal to the oracle-driven host code row for row"""
    _map_bsn_at_size(tmp_path, monkeypatch, 10000, 0, n_genomes, None, (3, 40), 1500)


def test_map_bsn_10k_exemplars_x_2000_genomes_through_the_worker_pool(tmp_path, monkeypatch):
    """BASELINE configs[4]'s genome count: 2 000 genomes (4.3 Gnt) against 10 000 exemplars through get_map_bsn with the reference's pool of
    workers (PEPPAN.py:907-923) as eight processes that share the GPU; every planted allele found, stores consistent, sampled genomes searched
    once more in this process and held to the oracle-driven host code row for row"""
    # (identity floor 0.90: among 4 M planted alleles a 120-base gene now and then collects seven substitutions at the nominal 2 %)
    _map_bsn_at_size(tmp_path, monkeypatch, 10000, 0, 2000, None, (7, 1234), 1500, min_iden4=9000, workers=8)


@pytest.mark.parametrize('n_genomes', [32, 128])
def test_map_bsn_50k_exemplars_x_genomes(tmp_path, monkeypatch, n_genomes):
    """BASELINE configs[4] mapping stage at the size one GPU holds: 50 000 exemplar genes (log-normal lengths, 45 Mnt) against 32 and 128 genomes of a
    50 000-gene pan-genome (about 6 500 genes / 7 Mb per genome: synth.PAN_GENOME_PRESENCE) through get_map_bsn (PEPPAN.py:759-772, 907-989);
    every planted allele found, stores consistent, one sampled genome equal to the oracle-driven host code row for row"""
    from peppan_amd import synth
    # (identity floor 0.90 here: with 210 000 planted alleles a 147-base gene now and then collects eight substitutions at the nominal 2 %)
    _map_bsn_at_size(tmp_path, monkeypatch, 50000, 0, n_genomes, synth.PAN_GENOME_PRESENCE, (5,), 5000, min_iden4=9000)


def _map_bsn_at_size(tmp_path, monkeypatch, n_genes, gene_len, n_genomes, presence, sample, planted_per_genome, min_iden4=9500, workers=None):
    from peppan_amd import mapbsn, synth, uberBlast as UB
    from oracle_context import OracleContext
    monkeypatch.chdir(tmp_path)
    names, seqs = synth.make_genes(n_genes, gene_len, seed=355)
    with open('m.clust.exemplar', 'w') as f:
        for i, s in enumerate(seqs):
            f.write('>%d\n%s\n' % (i, s.decode()))
    worlds = synth.make_genomes(seqs, n_genomes, seed=355, presence=presence)
    genomes = {}
    with mapbsn.MapBsn('m.old_prediction.npz', 'w') as op:
        for g, (gname, contig, ann) in enumerate(worlds):
            genomes[100000 + g] = [900000 + g, contig.decode()]
            op.save(100000 + g, np.array([[k, s, e, st, 1] for k, s, e, st in ann[::2]], dtype=object))
    np.save('m.self_bsn.npy', np.array([[0, 1, 9000], [4, 5, -2]], dtype=int))
    params = dict(noDiamond=False, match_identity=0.65, match_frag_len=50, match_frag_prop=0.25, link_gap=600, link_diff=1.5, gtable=11,
                  match_len=250., match_len1=100., match_len2=400., match_prop=0.5, match_prop1=0.8, match_prop2=0.4)
    seen = {}

    def search(prefix, clust, jobs, p):                      # the product's batched search, with the per-genome tables recorded for the sample
        for job, res in zip(jobs, mapbsn._gpu_search(prefix, clust, jobs, p, genomes_per_batch=32)):
            if job[0] in sample:
                seen[job[0]] = (res[0].to_rows(cigar='str'), res[1].copy())      # (the batch search hands over numeric HitTables)
            yield res
    fn = ['s.%s.npz' % x for x in ('tab', 'seq', 'mat', 'conflicts')]
    t0 = time.perf_counter()
    with contextlib.redirect_stderr(io.StringIO()):
        with mapbsn.MapBsn(fn[0], 'w') as c0, mapbsn.MapBsn(fn[1], 'w') as c1, mapbsn.MapBsn(fn[2], 'w') as c2, mapbsn.MapBsn(fn[3], 'w') as c3:
            if workers:
                # (the pool's processes run the product's batched search themselves; the sample is searched once more here, below)
                mapbsn.get_map_bsn('m', 'm.clust.exemplar', genomes, 'm.self_bsn.npy', 'm.old_prediction.npz', c0, c1, c2, c3, True, params, workers=workers)
            else:
                mapbsn.get_map_bsn('m', 'm.clust.exemplar', genomes, 'm.self_bsn.npy', 'm.old_prediction.npz', c0, c1, c2, c3, True, params, search=search)
    dt = time.perf_counter() - t0
    if workers:
        some = [(g, 900000 + g, [[100000 + g, genomes[100000 + g][1]]]) for g in sample]
        with contextlib.redirect_stderr(io.StringIO()):
            for job, res in zip(some, mapbsn._gpu_search('m', 'm.clust.exemplar', some, params, genomes_per_batch=len(some))):
                seen[job[0]] = (res[0].to_rows(cigar='str'), res[1].copy())
    print('get_map_bsn: %d genomes in %.1f s = %.1f genomes/s' % (n_genomes, dt, n_genomes / dt))      # (a figure, not a condition: a slow host must not turn a parity suite red)
    with mapbsn.MapBsn(fn[0]) as c:
        tab = np.vstack([c.get(k) for k in c.keys()])
    with mapbsn.MapBsn(fn[2]) as c:
        n_mat = sum(len(c.get(k)) for k in c.keys())
    # one row per group, group ids dense, one hit-row block per group
    assert tab.shape[0] == n_mat and sorted(tab.T[5].tolist()) == list(range(tab.shape[0]))
    # every planted allele (<= 2 % substitutions, full length) is found in its genome with identity >= 0.95
    found = {(int(r[0]), int(r[1])) for r in tab if r[3] >= min_iden4}
    planted = 0
    for g, (gname, contig, ann) in enumerate(worlds):
        for k, s, e, strand in ann:
            planted += 1
            assert (k, 900000 + g) in found, (g, k)
    assert planted > n_genomes * planted_per_genome
    # the sampled genomes: whole tables equal to the same host code over the CPU oracle (one uberBlast call per genome)
    octx = OracleContext()
    monkeypatch.setattr(UB, 'get_context', lambda device=None: octx)
    jobs = {i: [[g, s[1]]] for i, (g, s) in enumerate(genomes.items())}           # get_map_bsn's job ids: one taxon per genome, in this order
    for id in sample:
        gfile = mapbsn._write_genome('o', id, jobs[id])
        with contextlib.redirect_stderr(io.StringIO()):
            o_tab, o_ovl = UB.uberBlast(['-r', gfile] + mapbsn._map_argv('m.clust.exemplar', params))
        _cmp_tables(seen[id][0], o_tab)
        assert seen[id][1].tolist() == o_ovl.tolist()
        assert o_tab.shape[0] > 3 * planted_per_genome


@pytest.mark.parametrize('n_base,copies', [(4000, 250), (10000, 500), (50000, 400), (50000, 1000)])
def test_front_end_gene_instances_at_size(ctx, tmp_path, monkeypatch, n_base, copies):
    """writeGenes (K13 sha1 + duplicate collapse) and the 11-level iterClust (K9) on 1 M instances of 4 000 genes, at BASELINE
    configs[2]'s full size - 5 M instances of 10 000 genes (10 000 genes x 500 genomes, 4.45 Gnt) -, on 20 M and on 50 M instances of 50 000 genes
    (18 / 45 Gnt: half of configs[4]'s 100 M instances in the suite; all 100 M: tools/front_end_scale.py, profiles/r05_front_end_scale.txt;
    PEPPAN.py:1023-1039, 1777-1792)"""
    from peppan_amd import pipeline as PL
    if n_base * copies > 5000000:
        import psutil
        limit = psutil.virtual_memory().available
        try:
            limit = min(limit, int(open('/sys/fs/cgroup/memory.max').read()))       # (what the container grants, when that is less than the machine's)
        except (OSError, ValueError):
            pass
        need = 5e9 * n_base * copies / 1e6 / 2.                                     # (measured: 21 GB peak at 20 M instances; twice that asked for)
        if limit < need:
            pytest.skip('%d M gene instances need about %.0f GB of host memory; %.0f GB available' % (n_base * copies // 1000000, need / 1e9, limit / 1e9))
    monkeypatch.chdir(tmp_path)
    t0 = time.perf_counter()
    from peppan_amd import synth
    seqs = synth.make_instances(n_base, copies, seed=8)
    n = len(seqs)
    assert n == n_base * copies
    t1 = time.perf_counter()
    hashes = PL.gene_hashes(seqs, ctx=ctx)
    t2 = time.perf_counter()
    import hashlib
    for i in (0, 1, 777777, n - 1):
        assert hashes[i] == int(hashlib.sha1(seqs[i].encode()).hexdigest(), 16)
    genes = {i: ['f', '', 0, 0, '+', hashes[i], seqs[i]] for i in range(n)}
    prio = {i: [i % 7, -len(seqs[i]), hashes[i]] for i in range(n)}
    fn, groups = PL.writeGenes('big.genes', genes, prio, ctx=ctx)
    t3 = time.perf_counter()
    n_unique = sum(1 for line in open(fn) if line.startswith('>'))
    assert n_unique + len(groups) == n and n_base <= n_unique <= 7 * 5 * n_base     # identical alleles collapse inside a length run: at most 5 alleles x 7 file ranks per gene (PEPPAN.py:1032-1033)
    with contextlib.redirect_stderr(io.StringIO()):
        ex = PL.iterClust('big', fn, groups, dict(identity=0.9, coverage=0.8, n_thread=1, translate=False))
    t4 = time.perf_counter()
    n_ex = sum(1 for line in open(ex) if line.startswith('>'))
    clu = np.load('big.clust.npy')
    assert 0.6 * n_base <= n_ex <= n_base                     # the alleles of a gene end up under one exemplar; of the four members of a family the two closest (0 / 5 % substitutions) merge at 0.9
    assert clu.shape[0] >= n - n_ex - 12 * 11                 # (every level loses the first line of its table, PEPPAN.py:1786)
    rss = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6
    print('%d instances: generate %.1f s, sha1 %.1f s, writeGenes %.1f s, iterClust %.1f s; %d unique, %d exemplars; peak RSS %.1f GB'
          % (n, t1 - t0, t2 - t1, t3 - t2, t4 - t3, n_unique, n_ex, rss))
    assert rss < 40 * max(1.0, n / 5e6)                       # memory is a property of the code; the times above are printed, not asserted


def test_all_vs_all_50k_bit_exact_vs_oracle(ctx):
    """BASELINE configs[4] search stage: 50 000 genes x 1 002 nt all-vs-all on one GPU; every field of every hit and the CIGAR arena equal the
    CPU oracle's (OpenMP, all host cores: about a minute and a half).  Division of labour: the oracle is fed the proteins K1 made on the GPU
    (ctx.query_aa() / ctx.target_aa()), so at this size K1 - translation, frame choice, chunking - is held to itself; K1 against the oracle and the
    reference's golden vectors is test_gpu_parity.py's business (test_k1_*, G1 / G2 on the GPU) at sizes the numpy translation finishes in seconds."""
    from peppan_amd import _native as N, synth
    from oracle import oracle as O
    names, seqs = synth.make_genes(50000, 1002, seed=355)
    order = sorted(range(len(names)), key=lambda i: names[i])
    nts = [seqs[i] for i in order]
    ctx.set_query_nt(nts, 11)
    ctx.set_ref_nt(nts, 6, 11)
    p = N.default_params(45., 25., 10, 5)
    ctx.search(p)
    t0 = time.perf_counter()
    gh, gc, st = ctx.search(p)
    t_gpu = time.perf_counter() - t0
    qa, qo = ctx.query_aa()
    ta, to = ctx.target_aa()
    q_aa = [qa[qo[i]:qo[i + 1]] for i in range(len(qo) - 1)]
    t_aa = [ta[to[i]:to[i + 1]] for i in range(len(to) - 1)]
    O.lib().oracle_set_threads(O.granted_cpus())
    t0 = time.perf_counter()
    oh, oc, ost = O.search(q_aa, t_aa, O.default_params(45., 25., 10, 5))
    t_cpu = time.perf_counter() - t0
    print('50k x 50k: GPU search %.1f ms, oracle %.1f s, %d candidates, %d hits' % (t_gpu * 1e3, t_cpu, st['candidates'], len(gh)))
    assert len(gh) == len(oh) > 150000
    for f in ('q', 't', 'q_start', 'q_end', 't_start', 't_end', 'score', 'nm', 'n_ident', 'aln_len', 'cigar_runs', 'bin', 'cells'):
        assert np.array_equal(gh[f], oh[f]), f
    assert np.array_equal(gc, oc)
    for k in ('candidates', 'pairs', 'cells', 'tracebacks'):
        assert st[k] == ost[k], k


def test_chromosome_longer_than_the_sequence_limit(tmp_path, monkeypatch):
    """the nucleotide tool on a 10.6 Mbp contig (> PEP_MAX_SEQ_LEN = 8.39 Mbp: runBlast searches it as three overlapping windows, nothing
    patched): 300 genes planted on both strands, some of them right across the window boundaries at 4 194 304 and 8 388 608, all come
    back at their coordinates; the same genes on a second, short contig are found as well"""
    from peppan_amd import _native as N, synth, uberBlast as UB
    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(99)
    names, seqs = synth.make_genes(400, 0, seed=41)
    genes = [s for s in seqs if 300 <= len(s) <= 3000][:300]
    B = np.frombuffer(b'ACGT', dtype=np.uint8)
    comp = np.zeros(256, dtype=np.uint8)
    comp[list(b'ACGT')] = list(b'TGCA')
    total = 10600000
    assert total > N.MAX_SEQ_LEN
    chrom = B[rng.integers(0, 4, total)]
    # planted (gene, start, strand): one gene right across every window boundary - the windows of the forward strand meet at 4 194 304 and
    # 8 388 608, those of the reverse strand at the same offsets from the other end - with the midpoint 100 / 500 nt to either side,
    # neighbours a few kb away on both sides (inside the halos), the rest spread out; no two planted genes overlap
    H = UB._TILE_HOME
    edges = [(H, 0, -100), (2 * H, 0, 500), (total - H, 1, 100), (total - 2 * H, 1, -500)]
    planted, gi = [], 0
    for b, strand, shift in edges:
        for off in (0, -4000, 4000):
            g = genes[gi]
            planted.append((gi, b + (shift if off == 0 else off) - len(g) // 2, strand))
            gi += 1
    for pos in np.linspace(50000, total - 50000, len(genes) - gi + 60).astype(int):
        if gi >= len(genes):
            break
        if any(abs(int(pos) - b) < 20000 for b, _, _ in edges):
            continue
        planted.append((gi, int(pos), gi % 2))
        gi += 1
    ann = []
    for gi_, pos, strand in planted:
        s = np.frombuffer(genes[gi_], dtype=np.uint8).copy()
        m = rng.random(s.size) < 0.02
        s[m] = B[rng.integers(0, 4, int(m.sum()))]
        chrom[pos:pos + s.size] = comp[s[::-1]] if strand else s
        ann.append((gi_, pos + 1, pos + s.size, strand))
    with open('q.fa', 'w') as f:
        for i in range(len(planted)):
            f.write('>%d\n%s\n' % (i, genes[i].decode()))
    small = np.concatenate([chrom[p0 - 1:p1] for _, p0, p1, _ in ann[:20]])
    with open('r.fa', 'w') as f:
        f.write('>chr\n%s\n>small\n%s\n' % (chrom.tobytes().decode(), small.tobytes().decode()))
    argv = '-r r.fa -q q.fa --blastn -s 1 --min_id 0.6 --min_cov 50 --min_ratio 0.2 -e 0,3 -f -m'.split()
    t0 = time.perf_counter()
    with contextlib.redirect_stderr(io.StringIO()):
        tab = UB.uberBlast(argv)
    print('10.6 Mbp chromosome, %d queries: %.2f s, %d rows' % (len(planted), time.perf_counter() - t0, tab.shape[0]))
    on_chr = {}
    for r in tab:
        if r[1] == 'chr' and r[2] >= 0.95:
            on_chr.setdefault(int(r[0]), []).append((min(r[8], r[9]), max(r[8], r[9]), r[8] > r[9], r[6], r[7]))
    for gi_, p0, p1, strand in ann:
        hits = on_chr.get(gi_, [])
        assert any(abs(lo - p0) <= 3 and abs(hi - p1) <= 3 and rev == bool(strand) for lo, hi, rev, qs, qe in hits), (gi_, p0, p1, strand, hits[:3])
    assert sum(1 for r in tab if r[1] == 'small' and r[2] >= 0.95) >= 20
    # no duplicates from the overlapping halos: one row per (query, locus)
    for gi_, hits in on_chr.items():
        assert len({(lo, hi) for lo, hi, rev, qs, qe in hits}) == len(hits)
    # the same reference handed over in memory as ASCII bytes (what the mapping workers pass: no FASTA file, no str) through the batch entry point
    with contextlib.redirect_stderr(io.StringIO()):
        (batch,), = [UB.uberBlastBatch([[('chr', chrom.tobytes()), ('small', small.tobytes())]], argv[2:])]
    assert batch.tolist() == tab.tolist()


def test_example_genomes_full_pipeline_vs_oracle(ctx, tmp_path, monkeypatch):
    """BASELINE configs[0] / [1] on REAL genes at full size (golden G17: every CDS of the reference's four example genomes, 8 441 unique genes
    after its own writeGenes): sha1 (K13) == the reference's codes, writeGenes == its file and duplicate pairs, then the whole hot path
    iterClust (K9) -> get_similar_pairs (K1..K8 x 2 tools, K7, K14) -> get_gene_group / K10 on the GPU == the same host code over the CPU oracle"""
    import gzip, shutil
    from conftest import load_golden
    from peppan_amd import uberBlast as UB, pipeline as PL, clust as CL
    from oracle import oracle as O
    from oracle_context import OracleContext
    g = load_golden('g17_examples.json')
    with gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'g17_examples_genes.fa.gz'), 'rt') as f:
        text = f.read()
    seqs = {}
    for rec in text.split('>')[1:]:
        name, seq = rec.split('\n', 1)
        seqs[int(name)] = seq.strip()
    code = {name: int(c, 16) for name, length, c, rank in g['instances']}
    ids = sorted(seqs)
    assert len(ids) == 8441 and PL.gene_hashes([seqs[i] for i in ids], ctx=ctx) == [code[i] for i in ids]
    genes = {name: ['f', '', 0, 0, '+', int(c, 16), seqs.get(name, 'N' * length)] for name, length, c, rank in g['instances']}
    prio_all = {name: [rank, -length, int(c, 16)] for name, length, c, rank in g['instances']}
    monkeypatch.chdir(tmp_path)
    fn, groups = PL.writeGenes('ex.genes', genes, prio_all, ctx=ctx)
    assert open(fn).read() == text and groups == g['groups']
    prio = {int(k): [v[0], v[1], int(v[2])] for k, v in g['priority'].items()}
    params = dict(noDiamond=False, match_identity=0.5, match_frag_len=50, n_thread=2, match_frag_prop=0.25, gtable=11,
                  clust_identity=0.9, clust_match_prop=0.8, incompleteCDS='', match_len=250., match_len1=100., match_len2=400.,
                  match_prop=0.5, match_prop1=0.8, match_prop2=0.4)

    def oracle_fn(fasta, identity, coverage, n_thread):
        recs = CL.readFasta(fasta)
        rep, _ = O.linclust([O.nt_codes(s) for _, s in recs], float(identity), float(coverage))
        return [(recs[r][0], recs[i][0]) for i, r in enumerate(rep.tolist())]

    results, took = {}, {}
    for tag in ('gpu', 'ora'):
        d = tmp_path / tag
        d.mkdir()
        monkeypatch.chdir(d)
        shutil.copy(str(tmp_path / 'ex.genes'), 'p.genes')
        if tag == 'ora':
            octx = OracleContext()
            monkeypatch.setattr(UB, 'get_context', lambda device=None: octx)
            monkeypatch.setattr(PL, 'get_context', lambda device=None: octx)
        t0 = time.perf_counter()
        with contextlib.redirect_stderr(io.StringIO()):
            ex = PL.iterClust('p', 'p.genes', [list(x) for x in groups], dict(identity=0.9, coverage=0.8, n_thread=2, translate=False,
                                                                               cluster_fn=oracle_fn if tag == 'ora' else None))
            t1 = time.perf_counter()
            pairs = PL.get_similar_pairs(ex, prio, dict(params, clust=ex))
        t2 = time.perf_counter()
        np.save('p.self_bsn.npy', pairs)
        grp = PL.get_gene_group(ex, 'p.self_bsn.npy')
        took[tag] = (t1 - t0, t2 - t1)
        results[tag] = (open(ex).read(), np.load('p.clust.npy').tolist(), pairs.tolist(), [[int(k), [int(x) for x in v]] for k, v in grp.items()])
    print('example genomes, 8 441 genes: iterClust %.1f s, get_similar_pairs %.2f s on the GPU; %.1f s / %.1f s over the CPU oracle; %d pairs, %d groups'
          % (took['gpu'] + took['ora'] + (len(results['gpu'][2]), len(results['gpu'][3]))))
    assert results['gpu'] == results['ora']
    assert len(results['gpu'][2]) > 500 and len(results['gpu'][3]) > 200


def test_linclust_1m_sequences_bit_exact_vs_oracle(ctx):
    """K9 at the size the front end meets it: 1 M gene instances (4 000 genes x 250 alleles, 0.9 Gnt) through pep_linclust and through
    oracle_linclust (OpenMP over the sequences since round 5) - the same representative for every sequence and the same counts of selected
    k-mers, verified pairs and accepted edges.  (The at-size front-end tests above hold the 11-level schedule to properties; this holds one
    level to the oracle.)"""
    from peppan_amd import synth
    from oracle import oracle as O
    seqs = synth.make_instances(4000, 250, seed=8)
    lut = np.full(256, 4, dtype=np.uint8)
    for ch, v in zip('ACGT', (0, 1, 2, 3)):
        lut[ord(ch)] = v
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum(np.fromiter(map(len, seqs), dtype=np.int64, count=len(seqs)))
    codes = lut[np.frombuffer(''.join(seqs).encode('ascii'), dtype=np.uint8)]
    t0 = time.perf_counter()
    rep, st = ctx.linclust((codes, off), 0.97, 0.8)
    t1 = time.perf_counter()
    import ctypes as C
    o_rep = np.zeros(len(seqs), dtype=np.uint32)
    o_stats = (C.c_uint64 * 3)()
    O.lib().oracle_linclust(codes.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), C.c_uint32(len(seqs)), C.c_int(4), C.c_int(17), C.c_int(20),
                            C.c_double(0.97), C.c_double(0.8), o_rep.ctypes.data_as(C.c_void_p), o_stats)
    t2 = time.perf_counter()
    print('linclust of %d sequences: GPU %.2f s, oracle %.1f s; %d clusters' % (len(seqs), t1 - t0, t2 - t1, len(np.unique(rep))))
    assert (st['selected'], st['verified'], st['accepted']) == (int(o_stats[0]), int(o_stats[1]), int(o_stats[2]))
    assert np.array_equal(rep, o_rep)
    assert 3000 <= len(np.unique(rep)) <= 6000                 # (alleles under their gene - a short gene with three substitutions stays apart at 0.97; of a family's four genes the closest two are 5 % apart: not merged)
