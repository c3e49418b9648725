#!/usr/bin/env python3
"""Generate golden input/output vectors by IMPORTING the reference's Python.

Run in the build container only (needs /root/reference, which does not exist on
the GPU box):

    python tests/golden/make_golden.py

The script builds the shim described in SURVEY.md section 8c in a temp dir
(no-op / scripted `diamond blastn makeblastdb mmseqs` on PATH, stub `numba`
and `ete3` packages), imports `uberBlast`, `clust`, `configure` and `PEPPAN`
from /root/reference, drives every hot-path function with seeded synthetic
inputs and writes the (input, expected output) pairs as JSON next to this file.
Only DATA is written - none of the reference's source text.

Fixture index:
  g01_tables.json        blosum62, gtable, nucEncoder   (configure.py:49-87, uberBlast.py:270-272)
  g01_transeq.json       configure.transeq              (configure.py:160-194)
  g02_rundiamond.json    FASTA text runDiamond hands to diamond (uberBlast.py:525-549)
  g03_parsediamond.json  parseDiamond                   (uberBlast.py:16-70)
  g04_poolblast.json     poolBlast/parseBlast/getCIGAR  (uberBlast.py:274-320)
  g05_rescore.json       cigar2score / reScore          (uberBlast.py:221-269, 397-415)
  g06_fixend.json        fixEnd                         (uberBlast.py:462-480)
  g07_filters.json       ovlFilter/_linearMerge/returnOverlap (uberBlast.py:73-218, 378-460)
  g18_filters_random.json.gz  200 random mapping tables through the reference's ovlFilter / linearMerge (uberBlast.py:100-218, 417-460)
  g08_run.json           RunBlast.run / uberBlast end to end with canned aligner output
  g09_clust.json         getClust / iterClust           (clust.py:34-111, PEPPAN.py:1777-1792)
  g10_pairs.json         get_similar_pairs              (PEPPAN.py:194-294)
  g11_groups.json        get_gene_group                 (PEPPAN.py:1590-1609)
  g12_writegenes.json    writeGenes                     (PEPPAN.py:1023-1039)
  g13_readers.json       readFasta / readFastq / uopen  (configure.py:90-150, clust.py:7-18)
  g14_mapbsn.json        iter_map_bsn / compare_prediction / decodeSeq (PEPPAN.py:318-324, 759-905)
  g15_getmapbsn.json     get_map_bsn -> MapBsn stores   (PEPPAN.py:27-114, 907-989)
  g16_real.json (+ g16_real_genes.fa.gz, g16_real_contig.fa.gz)  real genes of the reference's examples/ through its own front end
  g17_examples.json (+ g17_examples_genes.fa.gz)  BASELINE configs[0] at full size: ALL CDS of the four example GFFs through the reference's
                         readGFF -> encodeNames -> load_priority -> writeGenes (PEPPAN.py:117-191, 746-751, 1023-1039, 1766-1775, 1844-1849):
                         the 8 441 unique genes (7.80 Mnt) of its <prefix>.genes with their priorities; instance hashes and duplicate groups.
                         (SURVEY.md 8a quotes 11 696 genes / 10.65 Mnt for this set from an earlier probe; the reference's writeGenes, run here, collapses
                         the 19 490 instances to 8 441 - there are exactly 8 441 distinct (length, sha1) among them)
"""
import json, os, sys, stat, tempfile, shutil, io, contextlib, copy

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'

FAKE_TOOL = r'''#!/usr/bin/env python3
# scripted stand-in for diamond / blastn / makeblastdb / mmseqs used ONLY while
# generating golden vectors: it records the files it is given and replays canned output.
import sys, os, shutil, json
tool = os.path.basename(sys.argv[0]); a = sys.argv[1:]
d = os.environ.get('FAKE_DIR')
if d is None: sys.exit(0)
def opt(name):
    return a[a.index(name) + 1]
log = os.path.join(d, 'calls.jsonl')
with open(log, 'a') as f: f.write(json.dumps([tool] + a) + '\n')
if tool == 'diamond' and a and a[0] == 'blastp':
    db, out = opt('--db'), opt('--out')
    sid = db.rsplit('.', 1)[1]
    shutil.copy(db, os.path.join(d, 'refAA.' + sid))
    shutil.copy(opt('--query'), os.path.join(d, 'qryAA'))
    src = os.path.join(d, 'sam.' + sid)
    if os.path.exists(src): shutil.copy(src, out)
    else: open(out, 'w').close()
elif tool == 'blastn':
    q, out = opt('-query'), opt('-out')
    if os.path.abspath(q) != os.path.abspath(os.path.join(d, os.path.basename(q))): shutil.copy(q, os.path.join(d, os.path.basename(q)))
    src = os.path.join(d, 'bsn.' + q.rsplit('.', 1)[1])
    if os.path.exists(src): shutil.copy(src, out)
    else: open(out, 'w').close()
elif tool == 'makeblastdb':
    shutil.copy(opt('-in'), os.path.join(d, 'refNA'))
elif tool == 'mmseqs' and a and a[0] == 'createtsv':
    cnt_f = os.path.join(d, 'mm.count')
    n = int(open(cnt_f).read()) if os.path.exists(cnt_f) else 0
    open(cnt_f, 'w').write(str(n + 1))
    shutil.copy(os.path.join(d, 'mm.input.%d' % n), os.path.join(d, 'mm.seen.%d' % n)) if os.path.exists(os.path.join(d, 'mm.input.%d' % n)) else None
    shutil.copy(os.path.join(d, 'clust.tab.%d' % n), a[4])
elif tool == 'mmseqs' and a and a[0] == 'createdb':
    cnt_f = os.path.join(d, 'mm.count')
    n = int(open(cnt_f).read()) if os.path.exists(cnt_f) else 0
    shutil.copy(a[1], os.path.join(d, 'mm.input.%d' % n))
sys.exit(0)
'''


def build_shim():
    root = tempfile.mkdtemp(prefix='peppan_shim_')
    os.makedirs(os.path.join(root, 'bin'))
    for t in ('mmseqs', 'makeblastdb', 'diamond', 'blastn'):
        p = os.path.join(root, 'bin', t)
        with open(p, 'w') as f:
            f.write(FAKE_TOOL)
        os.chmod(p, os.stat(p).st_mode | stat.S_IEXEC | stat.S_IXGRP | stat.S_IXOTH)
    for pkg, body in (('numba', 'def jit(*a, **k):\n    if len(a) == 1 and callable(a[0]) and not k:\n        return a[0]\n    return lambda f: f\n'),
                      ('ete3', 'class Tree(object):\n    pass\n')):
        os.makedirs(os.path.join(root, 'py', pkg))
        with open(os.path.join(root, 'py', pkg, '__init__.py'), 'w') as f:
            f.write(body)
    os.makedirs(os.path.join(root, 'cwd'))
    return root


SHIM = build_shim()
os.environ['PATH'] = os.path.join(SHIM, 'bin') + os.pathsep + os.environ['PATH']
sys.path[:0] = [os.path.join(SHIM, 'py'), os.path.join(REF, 'modules'), REF]
os.chdir(os.path.join(SHIM, 'cwd'))

import numpy as np
import pandas as pd
if not hasattr(np.lib.npyio, 'format'):   # numpy >= 2 dropped this alias of np.lib.format; MapBsn (PEPPAN.py:40, 89) spells it the old way
    np.lib.npyio.format = np.lib.format
import configure, uberBlast, clust  # noqa: E402  (the reference, flat imports)
import PEPPAN as PEP                 # noqa: E402


def jsonable(x):
    if isinstance(x, np.ndarray):
        return [jsonable(v) for v in x.tolist()]
    if isinstance(x, (list, tuple)):
        return [jsonable(v) for v in x]
    if isinstance(x, dict):
        return {str(k): jsonable(v) for k, v in x.items()}
    if isinstance(x, (np.integer,)):
        return int(x)
    if isinstance(x, (np.floating,)):
        return float(x)
    if isinstance(x, (np.bool_,)):
        return bool(x)
    return x


def dump(name, obj):
    with open(os.path.join(HERE, name), 'w') as f:
        json.dump(jsonable(obj), f, separators=(',', ':'))
    print('wrote', name, os.path.getsize(os.path.join(HERE, name)), 'bytes')


def fake_dir():
    d = tempfile.mkdtemp(prefix='fake_', dir=SHIM)
    os.environ['FAKE_DIR'] = d
    return d


# ----------------------------------------------------------------------------- sequence helpers
SENSE = [a + b + c for a in 'ACGT' for b in 'ACGT' for c in 'ACGT' if a + b + c not in ('TAA', 'TAG', 'TGA')]
COMP = {'A': 'T', 'C': 'G', 'G': 'C', 'T': 'A', 'N': 'N'}


def revcomp(s):
    return ''.join(COMP.get(c, 'N') for c in reversed(s))


def rand_cds(rng, ncodon):
    return 'ATG' + ''.join(SENSE[i] for i in rng.integers(0, len(SENSE), ncodon - 2)) + 'TAA'


def mutate_cds(rng, cds, sub=0.05, indel=0.01):
    """codon-level derivative of `cds` + the aa-level alignment ops (query=cds, ref=derivative)"""
    codons = [cds[i:i + 3] for i in range(0, len(cds), 3)]
    out, ops = [], []
    for ci, c in enumerate(codons):
        edge = ci < 2 or ci >= len(codons) - 2
        r = rng.random()
        if not edge and r < indel:               # codon deleted from the derivative: query-only => I
            ops.append('I')
            continue
        if not edge and r < 2 * indel:           # extra codon in the derivative: ref-only => D
            out.append(SENSE[rng.integers(0, len(SENSE))])
            ops.append('D')
        cc = list(c)
        if not edge:
            for k in range(3):
                if rng.random() < sub:
                    cc[k] = 'ACGT'[rng.integers(0, 4)]
            if ''.join(cc) in ('TAA', 'TAG', 'TGA'):
                cc = list(c)
        out.append(''.join(cc))
        ops.append('M')
    return ''.join(out), ops


def rle(ops):
    runs = []
    for o in ops:
        if runs and runs[-1][1] == o:
            runs[-1][0] += 1
        else:
            runs.append([1, o])
    return runs


def cigar_str(runs):
    return ''.join('%d%s' % (n, o) for n, o in runs)


# ----------------------------------------------------------------------------- G1 tables + transeq
def g01():
    dump('g01_tables.json', dict(
        blosum62=configure.blosum62.astype(int),
        gtable=uberBlast.gtable,
        nucEncoder={c: int(uberBlast.nucEncoder[ord(c)]) for c in 'ACGTNRYKM-acgtn'},
        baseConv={c: int(configure.baseConv[ord(c)]) for c in 'ACGTN-RY'},
    ))
    rng = np.random.default_rng(101)
    cases = []
    seqs = {
        's0': 'ATGAAATAGCC',
        's1': rand_cds(rng, 40),
        's2': rand_cds(rng, 33) + 'A',
        's3': rand_cds(rng, 21) + 'CG',
        's4': 'ATGNNNAAARTTTGA-CCTGA',
        's5': 'TGATGATGATGAAATGA',
        's6': 'AC',
        's7': 'A',
        's8': ''.join('ACGT'[i] for i in rng.integers(0, 4, 301)),
        's9': 'atgaaacccgggtttTAA',
    }
    for frame in ('F', 'R', '7', 1, '2', '1,4', 6):
        for table in (11, 4, None):
            out = configure.transeq(seqs, frame=frame, transl_table=table)
            cases.append(dict(frame=frame, table=table, out=out))
    # list input + markStarts (clust.py:42 uses frame='1', transl_table='starts')
    lst = [[k, v] for k, v in seqs.items()]
    cases.append(dict(frame='1', table='starts', list_input=True, out=configure.transeq(lst, frame='1', transl_table='starts')))
    cases.append(dict(frame=1, table=11, markStarts=True, out=configure.transeq(seqs, frame=1, transl_table=11, markStarts=True)))
    dump('g01_transeq.json', dict(seqs=seqs, cases=cases, rc={k: configure.rc(v) for k, v in seqs.items()}))


# ----------------------------------------------------------------------------- gene set used by G2..G8
def make_gene_world(seed=7, long_contig=True):
    """queries = genes (int names like PEPPAN), refs = genes or contigs carrying derivatives"""
    rng = np.random.default_rng(seed)
    genes, rel = {}, []
    gid = 100
    for fam in range(6):
        root = rand_cds(rng, int(rng.integers(50, 140)))
        genes[str(gid)] = root
        fam_ids = [gid]
        gid += 7
        for sub in (0.02, 0.12):
            der, ops = mutate_cds(rng, root, sub=sub, indel=0.015)
            genes[str(gid)] = der
            rel.append((str(fam_ids[0]), str(gid), ops))
            fam_ids.append(gid)
            gid += 13
    # a gene with ambiguous bases / not multiple of 3
    genes['9'] = 'ATGAAACCNGGGTTTACGATTTTGCAGGCATCCGATAAGTAA' + 'C'
    genes['10'] = rand_cds(rng, 45)[:-3] + 'TGGTGATT'
    return rng, genes, rel


def write_fasta(path, seqs, width=None):
    with open(path, 'w') as f:
        for n, s in seqs.items():
            f.write('>%s some description\n' % n)
            if width:
                for i in range(0, len(s), width):
                    f.write(s[i:i + width] + '\n')
            else:
                f.write(s + '\n')


def g02():
    rng, genes, rel = make_gene_world()
    # a long contig (> 3000 nt so frames exceed 1000 aa and get cut at stop codons)
    contig = ''.join('ACGT'[i] for i in rng.integers(0, 4, 700))
    for k in ('100', '107', '120', '140'):
        contig += genes[k] + ''.join('ACGT'[i] for i in rng.integers(0, 4, int(rng.integers(40, 90))))
        contig += revcomp(genes[k]) + ''.join('ACGT'[i] for i in rng.integers(0, 4, int(rng.integers(40, 90))))
    longorf = rand_cds(rng, 1300)
    contig += longorf + ''.join('ACGT'[i] for i in rng.integers(0, 4, 211))
    refs = dict(genes)
    refs['ctg:1'] = contig
    refs['ctg:2'] = longorf[:3003]
    out = {}
    for frames_name, method in (('7', 'runDiamond'), ('F', 'runDiamondSELF')):
        d = fake_dir()
        qf, rf = os.path.join(d, 'q.fa'), os.path.join(d, 'r.fa')
        write_fasta(qf, genes, 60)
        write_fasta(rf, refs, 70)
        rb = uberBlast.RunBlast()
        rb.min_id, rb.min_cov, rb.min_ratio, rb.table_id, rb.n_thread = 0.4, 40, 0.1, 11, 2
        from multiprocessing.pool import ThreadPool
        rb.pool = ThreadPool(2)
        rb.dirPath = tempfile.mkdtemp(prefix='NS_', dir='.')
        try:
            with contextlib.redirect_stderr(io.StringIO()):
                getattr(rb, method)(rf, qf)
        except ValueError:
            pass  # np.vstack([]) when the scripted diamond returns no hits (uberBlast.py:558)
        shutil.rmtree(rb.dirPath)
        calls = [json.loads(l) for l in open(os.path.join(d, 'calls.jsonl'))]
        out[frames_name] = dict(
            qryAA=open(os.path.join(d, 'qryAA')).read(),
            refAA=[open(os.path.join(d, 'refAA.%d' % i)).read() for i in range(5)],
            diamond_args=[[os.path.basename(x) if '/' in x else x for x in c] for c in calls if c[0] == 'diamond'],
        )
    dump('g02_rundiamond.json', dict(query=genes, ref=refs, table_id=11, min_id=0.4, min_ratio=0.1, n_thread=2, out=out))
    return genes, refs, rel


# ----------------------------------------------------------------------------- G3 parseDiamond
def craft_sam_records(rng, genes, refs, rel, chunks):
    """SAM (outfmt 101) lines as diamond would emit them for known codon-level alignments.
    chunks: {(refname, frame): [(offset, length), ...]} from the recorded refAA files."""
    lines = []
    trans_q = configure.transeq(genes, frame='F', transl_table=11)

    def q_frame(n):
        ss = trans_q[n]
        return min((len(s[:-1].split('X')), i, s) for i, s in enumerate(ss))[1] + 1

    def emit(qn, rn, rframe, r_aa_start, q_aa_start, runs, nm, score):
        ch = chunks[(rn, rframe)]
        off = [o for o, l in ch if o < r_aa_start <= o + l]
        if not off:
            return
        off = off[0]
        qm = sum(n for n, o in runs if o in 'MI')
        tags = ['AS:i:%d' % int(score * 0.4), 'NM:i:%d' % nm, 'ZL:i:%d' % 300, 'ZR:i:%d' % score,
                'ZE:f:1e-30', 'ZI:i:90', 'ZF:i:1', 'ZS:i:%d' % q_aa_start, 'MD:Z:10']
        lines.append('\t'.join(['%s:%d' % (qn, q_frame(qn)), '0', '%s:%d:%d' % (rn, rframe, off), str(r_aa_start - off),
                                '255', cigar_str(runs), '*', '0', '0', 'A' * qm, '*'] + tags))

    # full-length gene-vs-gene family hits (forward frame 1)
    for qn, rn, ops in rel:
        runs = rle(ops)
        nm = int(rng.integers(0, 12)) + sum(n for n, o in runs if o != 'M')
        emit(qn, rn, 1, 1, 1, runs, nm, int(rng.integers(150, 700)))
        # the reverse relation (swap I/D)
        sw = [[n, {'M': 'M', 'I': 'D', 'D': 'I'}[o]] for n, o in runs]
        emit(rn, qn, 1, 1, 1, sw, nm, int(rng.integers(150, 700)))
        # partial alignment starting inside both
        part = ops[5:-7]
        while part and part[0] != 'M':
            part = part[1:]
        while part and part[-1] != 'M':
            part = part[:-1]
        if len(part) > 20:
            qs = 1 + sum(1 for o in ops[:len(ops) - 7 - len(ops[5:-7]) + 0] if o in 'MI') if False else None
            # recompute starts by counting consumed residues before the retained window
            lead = ops[:ops.index('M', 5)] if 'M' in ops[5:] else ops[:5]
            k = len(ops) - len(ops[5:-7]) - 7  # == 5
            pre = ops[:5]
            extra = ops[5:-7]
            skip = len(extra) - len(''.join(extra).lstrip('ID'))
            pre = ops[:5 + skip]
            qs = 1 + sum(1 for o in pre if o in 'MI')
            rs = 1 + sum(1 for o in pre if o in 'MD')
            emit(qn, rn, 1, rs, qs, rle(part), int(rng.integers(0, 9)) + sum(1 for o in part if o != 'M'), int(rng.integers(60, 400)))
    return lines


def parse_chunks(ref_texts):
    chunks = {}
    for t in ref_texts:
        lines = t.strip().split('\n') if t.strip() else []
        for h, s in zip(lines[0::2], lines[1::2]):
            rn, fr, off = h[1:].rsplit(':', 2)
            chunks.setdefault((rn, int(fr)), []).append((int(off), len(s)))
    return chunks


def contig_hits(rng, genes, refs, chunks):
    """hits of genes placed inside ctg:1 on both strands -> exercises frames 1..6 and chunk offsets"""
    lines = []
    contig = refs['ctg:1']
    rl = len(contig)
    trans_q = configure.transeq(genes, frame='F', transl_table=11)
    for k in ('100', '107', '120', '140'):
        g = genes[k]
        naa = len(g) // 3
        ss = trans_q[k]
        qf = min((len(s[:-1].split('X')), i, s) for i, s in enumerate(ss))[1] + 1
        # forward copy
        o = contig.find(g)
        # reverse copy: position in revcomp(contig)
        o_rc = revcomp(contig).find(g)
        for strand, off in (('+', o), ('-', o_rc)):
            frame = off % 3 + (1 if strand == '+' else 4)
            aa0 = off // 3 + 1
            # trim a few residues either side so that fixEnd has something to do
            for (lt, rt) in ((0, 0), (1, 2), (3, 0)):
                runs = [[naa - lt - rt, 'M']]
                ch = chunks.get(('ctg:1', frame), [])
                pos = aa0 + lt
                offc = [oo for oo, l in ch if oo < pos <= oo + l]
                if not offc:
                    continue
                offc = offc[0]
                tags = ['AS:i:100', 'NM:i:%d' % int(rng.integers(0, 3)), 'ZL:i:%d' % 300, 'ZR:i:%d' % int(rng.integers(200, 900)),
                        'ZE:f:1e-30', 'ZI:i:90', 'ZF:i:1', 'ZS:i:%d' % (1 + lt), 'MD:Z:10']
                lines.append('\t'.join(['%s:%d' % (k, qf), '0', 'ctg:1:%d:%d' % (frame, offc), str(pos - offc),
                                        '255', cigar_str(runs), '*', '0', '0', 'A' * (naa - lt - rt), '*'] + tags))
    return lines


def g03(genes, refs, rel):
    g2 = json.load(open(os.path.join(HERE, 'g02_rundiamond.json')))
    chunks = parse_chunks(g2['out']['7']['refAA'])
    rng = np.random.default_rng(33)
    lines = craft_sam_records(rng, genes, refs, rel, chunks) + contig_hits(rng, genes, refs, chunks)
    header = ['@HD\tVN:1.5\tSO:query', '@PG\tPN:DIAMOND', '@mm\tBlastP', '@CO\tBlastP-like alignments']
    unaligned = '999:1\t4\t*\t0\t255\t*\t*\t0\t0\t*\t*'
    # regex-fallback variant: tags in a different order (positional lookups fail, uberBlast.py:35-36,41,56)
    alt = lines[0].split('\t')
    alt = alt[:11] + [alt[18], alt[14], alt[12], alt[11]] + ['ZZ:i:1'] * 5
    text = '\n'.join(header + lines + [unaligned, '\t'.join(alt)]) + '\n'
    cases = []
    for (min_id, min_cov, min_ratio) in ((0.4, 40., 0.1), (0.9, 200., 0.5), (0.0, 0., 0.)):
        d = fake_dir()
        fn = os.path.join(d, 'aaMatch.0')
        with open(fn, 'w') as f:
            f.write(text)
        r = uberBlast.parseDiamond([fn, refs, genes, min_id, min_cov, min_ratio])
        rows = np.load(r, allow_pickle=True) if r else []
        cases.append(dict(min_id=min_id, min_cov=min_cov, min_ratio=min_ratio, rows=rows))
    dump('g03_parsediamond.json', dict(sam=text, qlen={k: len(v) for k, v in genes.items()},
                                       rlen={k: len(v) for k, v in refs.items()}, cases=cases))
    return text


# ----------------------------------------------------------------------------- G4 poolBlast
def gapped_strings(q, r, ops_nt):
    qa, ra, qi, ri = [], [], 0, 0
    for n, o in ops_nt:
        if o == 'M':
            qa.append(q[qi:qi + n]); ra.append(r[ri:ri + n]); qi += n; ri += n
        elif o == 'I':
            qa.append(q[qi:qi + n]); ra.append('-' * n); qi += n
        else:
            qa.append('-' * n); ra.append(r[ri:ri + n]); ri += n
    return ''.join(qa), ''.join(ra)


def craft_blast_lines(rng, genes, refs, rel):
    lines = []
    for qn, rn, ops in rel:
        runs = [[3 * n, o] for n, o in rle(ops)]
        q, r = genes[qn], genes[rn]
        qa, ra = gapped_strings(q, r, runs)
        alen = len(qa)
        mism = sum(1 for a, b in zip(qa, ra) if a != b and a != '-' and b != '-')
        gaps = sum(1 for n, o in runs if o != 'M')
        ident = sum(1 for a, b in zip(qa, ra) if a == b)
        pid = '%.3f' % (100. * ident / alen)
        score = 2 * ident - 3 * mism - sum(6 + 2 * n for n, o in runs if o != 'M')
        lines.append('\t'.join(map(str, [qn, rn, pid, alen, mism, gaps, 1, len(q), 1, len(r), '1e-50', score, len(q), len(r), qa, ra])))
        # reverse-strand hit of the same pair against the contig copy
    contig = refs['ctg:1']
    for k in ('100', '107'):
        g = genes[k]
        o = contig.find(g)
        lines.append('\t'.join(map(str, [k, 'ctg:1', '100.000', len(g) - 4, 0, 0, 3, len(g) - 2, o + 3, o + len(g) - 2, '0.0', 2 * (len(g) - 4), len(g), len(contig), g[2:-2], g[2:-2]])))
        o2 = contig.find(revcomp(g))
        lines.append('\t'.join(map(str, [k, 'ctg:1', '100.000', len(g) - 5, 0, 0, 2, len(g) - 4, o2 + len(g) - 1, o2 + 5, '0.0', 2 * (len(g) - 5), len(g), len(contig), g[1:-4], g[1:-4]])))
    # a low identity and a short hit that the filters drop
    lines.append('\t'.join(map(str, ['100', '120', '35.000', 90, 50, 0, 1, 90, 1, 90, '1e-3', 20, len(genes['100']), len(genes['120']), 'A' * 90, 'C' * 90])))
    lines.append('\t'.join(map(str, ['100', '120', '95.000', 20, 1, 0, 1, 20, 1, 20, '1e-3', 30, len(genes['100']), len(genes['120']), 'A' * 20, 'A' * 20])))
    return lines


def g04(genes, refs, rel):
    rng = np.random.default_rng(44)
    lines = craft_blast_lines(rng, genes, refs, rel)
    text = '\n'.join(lines) + '\n'
    cases = []
    for (min_id, min_cov, min_ratio) in ((0.4, 40., 0.1), (0.95, 100., 0.9)):
        d = fake_dir()
        qf = os.path.join(d, 'qryNA.0')
        open(qf, 'w').write('>x\nACGT\n')
        open(os.path.join(d, 'bsn.0'), 'w').write(text)
        r = uberBlast.poolBlast([uberBlast.blastn, 'refDb', qf, min_id, min_cov, min_ratio])
        rows = np.load(r, allow_pickle=True) if r else []
        cases.append(dict(min_id=min_id, min_cov=min_cov, min_ratio=min_ratio, rows=rows))
    cig = [dict(ref=ra, qry=qa, cigar=uberBlast.getCIGAR((ra, qa))) for qa, ra in
           [('ACGT', 'ACGT'), ('AC-GT', 'ACTGT'), ('ACGGT', 'A--GT'), ('--AC', 'GGAC'), ('AC--', 'ACGG'), ('A-C-G', 'ATCGG')]]
    dump('g04_poolblast.json', dict(outfmt6=text, cases=cases, getCIGAR=cig))
    return text


# ----------------------------------------------------------------------------- G5/G6 rescore + fixEnd
def base_table(genes, refs, sam_text, bsn_text):
    """the un-rescored table RunBlast.run would hold after vstack + id column (uberBlast.py:353-354)"""
    d = fake_dir()
    fn = os.path.join(d, 'aaMatch.0')
    open(fn, 'w').write(sam_text)
    r = uberBlast.parseDiamond([fn, refs, genes, 0.3, 40., 0.1])
    a = np.load(r, allow_pickle=True)
    qf = os.path.join(d, 'qryNA.0')
    open(qf, 'w').write('>x\nACGT\n')
    open(os.path.join(d, 'bsn.0'), 'w').write(bsn_text)
    b = np.load(uberBlast.poolBlast([uberBlast.blastn, 'refDb', qf, 0.3, 40., 0.1]), allow_pickle=True)
    tab = np.vstack([b, a])
    tab = np.hstack([tab, np.arange(tab.shape[0], dtype=int)[:, np.newaxis]])
    return tab


def g05_g06(genes, refs, sam_text, bsn_text):
    d = fake_dir()
    qf, rf = os.path.join(d, 'q.fa'), os.path.join(d, 'r.fa')
    write_fasta(qf, genes, 60)
    write_fasta(rf, refs, 70)
    tab0 = base_table(genes, refs, sam_text, bsn_text)
    cases = []
    for mode in (1, 2, 3):
        for table_id in (11, 4):
            saved = uberBlast.gtable.copy()
            rb = uberBlast.RunBlast()
            tab = copy.deepcopy(tab0)
            out = rb.reScore(rf, qf, tab, mode, 0.5, table_id)
            uberBlast.gtable[:] = saved     # undo the table-4 side effect (uberBlast.py:223-224)
            cases.append(dict(mode=mode, table_id=table_id, min_id=0.5, rows=out))
    # raw cigar2score calls on tiny hand-made inputs
    enc = lambda s: uberBlast.nucEncoder[np.array(list(s)).view(configure.asc2int)]
    raw = []
    for cigar, r, q, frame in (
            ([[9, 'M']], 'ATGAAACCC', 'ATGAAGCCC', 1),
            ([[6, 'M'], [3, 'D'], [6, 'M']], 'ATGAAATTTCCCGGG', 'ATGAAACCCGGG', 1),
            ([[6, 'M'], [6, 'I'], [3, 'M']], 'ATGAAAGGG', 'ATGAAACCCTTTGGG', 2),
            ([[4, 'M'], [1, 'I'], [7, 'M'], [2, 'D'], [3, 'M']], 'ATGANACCGGTTTAAC', 'ATGACNACCGGTAAC', 3),
    ):
        for mode in (1, 2, 3):
            v = uberBlast.cigar2score([cigar, enc(r), enc(q), frame, mode, 6, 1, 11])
            raw.append(dict(cigar=cigar, r=r, q=q, frame=frame, mode=mode, out=[float(v[0]), float(v[1])]))
    dump('g05_rescore.json', dict(query=genes, ref=refs, table=tab0, cases=cases, raw=raw))

    fe = []
    for se, ee in ((3, 3), (0, 3), (6, 6), (0, 0)):
        rb = uberBlast.RunBlast()
        tab = copy.deepcopy(tab0)
        # shrink some alignments at the edges so both branches trigger
        rb.fixEnd(tab, se, ee)
        fe.append(dict(se=se, ee=ee, rows=tab))
    dump('g06_fixend.json', dict(table=tab0, cases=fe))
    return tab0


# ----------------------------------------------------------------------------- G7 filters
def genome_table(rng, n_gene=12, n_contig=2):
    """hand-made genome-mapping table: several genes hit contigs at overlapping / collinear loci"""
    rows = []
    rid = 0
    for g in range(n_gene):
        ql = int(rng.integers(300, 1500))
        for c in range(n_contig):
            sl = 20000
            nh = int(rng.integers(1, 5))
            base = int(rng.integers(100, sl - 3000))
            prev_qe = 0
            for h in range(nh):
                qs = prev_qe + int(rng.integers(1, 40)) if h else int(rng.integers(1, 30))
                qe = min(ql, qs + int(rng.integers(60, max(61, ql // nh))))
                if qe - qs < 50:
                    break
                span = qe - qs + int(rng.integers(-6, 7))
                strand = 1 if (g + c) % 3 else -1
                s0 = base + (qs if strand > 0 else (ql - qe)) + int(rng.integers(0, 50)) * (h > 0)
                ss, se = (s0, s0 + span) if strand > 0 else (s0 + span, s0)
                iden = round(float(rng.uniform(0.6, 1.0)), 3)
                score = float(int((qe - qs + 1) * (3 * iden - (1 - iden))))
                rows.append([str(g), 'ctg:%d' % c, iden, qe - qs + 1, 3, 0, qs, qe, ss, se, 0.0, score, ql, sl,
                             [[qe - qs + 1, 'M']], rid])
                rid += 1
                prev_qe = qe
        # a competing paralogous gene covering the same locus with lower score
        if g % 3 == 0 and rows:
            t = list(rows[-1])
            t = t[:]
            t[0] = str(g + 100); t[2] = round(t[2] - 0.05, 3); t[11] = t[11] - 30.; t[14] = copy.deepcopy(t[14]); t[15] = rid
            rows.append(t); rid += 1
    tab = np.empty([len(rows), 16], dtype=object)
    for i, r in enumerate(rows):
        for j, v in enumerate(r):
            tab[i, j] = v
    return tab


def crafted_filter_table():
    """rows built to walk every branch of ovlFilter (uberBlast.py:423-448) and the merge / edge logic of _linearMerge"""
    rows = []

    def add(q, r, iden, qs, qe, ss, se, score, ql=900, sl=20000):
        rows.append([q, r, iden, qe - qs + 1, 2, 0, qs, qe, ss, se, 0.0, float(score), ql, sl, [[qe - qs + 1, 'M']], len(rows)])
    add('500', 'c9', 0.95, 1, 900, 1000, 1899, 800)        # A
    add('500', 'c9', 0.90, 40, 860, 1050, 1870, 700)       # B inside A, weaker -> dropped (2nd branch)
    add('500', 'c9', 0.80, 1, 500, 3000, 3500, 300)        # C
    add('500', 'c9', 0.92, 1, 590, 3010, 3600, 500)        # D covers C, stronger -> C dropped (1st branch)
    add('500', 'c9', 0.85, 100, 200, 5000, 5100, 100)      # G short, contained in H on ref AND query (3rd branch, no-op + break)
    add('500', 'c9', 0.85, 50, 850, 5000, 5800, 100)       # H
    add('500', 'c9', 0.85, 1, 700, 7000, 7700, 90)         # I long
    add('500', 'c9', 0.85, 300, 400, 7100, 7200, 90)       # J contained in I on both axes (4th branch -> dropped)
    add('501', 'c9', 0.97, 1, 900, 1899, 1000, 850)        # reverse strand duplicates of A/B
    add('501', 'c9', 0.91, 30, 870, 1880, 1040, 650)
    add('502', 'c9', 0.9, 1, 300, 9000, 9299, 250)         # collinear fragments of one gene -> linearMerge chains them
    add('502', 'c9', 0.88, 320, 600, 9330, 9610, 230)
    add('502', 'c9', 0.86, 620, 900, 9640, 9920, 220)
    add('503', 'c7', 0.9, 1, 450, 19551, 20000, 400, 900, 20000)   # gene split over two contig ends (resolve_edges)
    add('503', 'c8', 0.9, 451, 900, 1, 450, 410, 900, 20000)
    add('504', 'c7', 0.9, 1, 450, 450, 1, 400, 900, 20000)         # same, reverse strand
    add('504', 'c8', 0.9, 451, 900, 20000, 19551, 410, 900, 20000)
    tab = np.empty([len(rows), 16], dtype=object)
    for i, r in enumerate(rows):
        for j, v in enumerate(r):
            tab[i, j] = v
    return tab


def g07():
    rng = np.random.default_rng(77)
    cases = []
    for rep in range(4):
        tab = genome_table(rng) if rep < 3 else crafted_filter_table()
        rb = uberBlast.RunBlast()
        f = rb.ovlFilter(copy.deepcopy(tab), [True, 0.9, 0.])
        f2 = rb.ovlFilter(copy.deepcopy(tab), [True, 0.5, 10.])
        m = rb.linearMerge(copy.deepcopy(f), [True, 600., 1.5])
        m2 = rb.linearMerge(copy.deepcopy(tab), [True, 300., 1.2])
        ov = rb.returnOverlap(copy.deepcopy(m), [True, 300, 0.6])
        ov2 = rb.returnOverlap(copy.deepcopy(tab), [True, 30, 0.1])
        cases.append(dict(table=tab, ovlFilter_09_0=f, ovlFilter_05_10=f2, linearMerge_600_15=m, linearMerge_300_12_raw=m2,
                          overlap_300_06=ov, overlap_30_01_raw=ov2))
    dump('g07_filters.json', dict(cases=cases))


# ----------------------------------------------------------------------------- G18 randomized -f / -m tables
G18_COLS = ('q', 'r', 'iden', 'qs', 'qe', 'ss', 'se', 'score', 'ql', 'sl')


def dense_filter_table(rng, n_gene, n_contig, per):
    """many fragments per gene and contig, close together: chains, joined chains, contig-edge pairs, more than eight rows per query,
    overlapping competitors.  Returns the ten defining columns; the other six of the 16-column row follow from them (expand_filter_table)."""
    cols = []
    for g in range(n_gene):
        ql = int(rng.integers(600, 3000))
        for c in range(n_contig):
            sl = int(rng.integers(4000, 9000))
            fwd = rng.random() < 0.5
            r_at = int(rng.integers(1, 400)) if rng.random() < 0.5 else int(rng.integers(sl // 2, sl - 1500))
            q_at = int(rng.integers(1, 40))
            for h in range(int(rng.integers(1, per + 1))):
                ln = int(rng.integers(60, 500))
                qs, qe = q_at, min(ql, q_at + ln)
                if qe - qs < 50:
                    break
                span = qe - qs + int(rng.integers(-6, 7))
                lo, hi = r_at, min(sl, r_at + span)
                if hi - lo < 40:
                    break
                if fwd:
                    rec = [qs, qe, lo, hi]
                else:
                    rec = [ql - qe + 1, ql - qs + 1, hi, lo]
                iden = round(float(rng.uniform(0.7, 1.0)), 3)
                n_q = rec[1] - rec[0] + 1
                score = float(int(n_q * (4 * iden - 1))) if rng.random() < 0.8 else float(int(n_q * 2))
                cols.append([str(g), 'c%d' % c, iden] + rec + [score, ql, sl])
                step = int(rng.integers(-30, 200))
                q_at, r_at = q_at + ln + step, r_at + span + step + int(rng.integers(-20, 20))
            if rng.random() < 0.3 and cols:                  # an overlapping competitor of the last hit
                t = list(cols[-1])
                t[7] = t[7] - int(rng.integers(-40, 40))
                if t[5] < t[6]:
                    t[5] += 3
                if t[5] != t[6]:
                    cols.append(t)
    return cols


def expand_filter_table(cols):
    tab = np.empty([len(cols), 16], dtype=object)
    for i, (q, r, iden, qs, qe, ss, se, score, ql, sl) in enumerate(cols):
        for j, v in enumerate([q, r, iden, qe - qs + 1, 3, 0, qs, qe, ss, se, 0.0, score, ql, sl, [[qe - qs + 1, 'M']], i]):
            tab[i, j] = v
    return tab


def g18():
    """200 random mapping tables through the reference's own ovlFilter (three settings) and linearMerge (three settings): what the host C++
    ports of -f / -m are held to beyond the crafted cases of G7.  Stored compactly: the ten defining columns of every input row; of every
    output the row ids in output order (the filters pass the other columns through) and, for -m, column 16 of every row."""
    rng = np.random.default_rng(1818)
    cases = []
    for k in range(200):
        cols = dense_filter_table(rng, int(rng.integers(3, 9)), int(rng.integers(1, 4)), (2, 5, 12, 20)[k % 4])
        tab = expand_filter_table(cols)
        rb = uberBlast.RunBlast()
        case = dict(cols=cols, ovl={}, merge={})
        for cov, delta in ((0.9, 0.), (0.5, 10.), (0.2, -5.)):
            f = rb.ovlFilter(copy.deepcopy(tab), [True, cov, delta])
            assert all(list(r[:15]) == list(tab[r[15]][:15]) for r in f)          # pass-through of every other column
            case['ovl']['%g_%g' % (cov, delta)] = [int(r[15]) for r in f]
        for gap, diff in ((600., 1.5), (300., 1.2), (2000., 3.0)):
            m = rb.linearMerge(copy.deepcopy(tab), [True, gap, diff])
            assert all(list(r[:15]) == list(tab[r[15]][:15]) for r in m)
            case['merge']['%g_%g' % (gap, diff)] = [[int(r[15]), r[16]] for r in m]
        cases.append(case)
    import gzip
    with gzip.GzipFile(os.path.join(HERE, 'g18_filters_random.json.gz'), 'wb', mtime=0) as f:
        f.write(json.dumps(jsonable(dict(columns=G18_COLS, cases=cases)), separators=(',', ':')).encode())
    print('wrote g18_filters_random.json.gz', os.path.getsize(os.path.join(HERE, 'g18_filters_random.json.gz')), 'bytes;',
          sum(len(c['cols']) for c in cases), 'rows;', sum(1 for c in cases for v in c['merge'].values() for r in v if len(r[1]) > 4), 'chained rows')


# ----------------------------------------------------------------------------- G8 RunBlast.run end to end
def g08(genes, refs, sam_text, bsn_text):
    out = []
    for name, argv_tail, ref_sel in (
            ('self', '--blastn --diamond -s 1 --min_id 0.45 --min_cov 50 -t 2 --min_ratio 0.25 -e 3,3 --gtable 11', None),
            ('self_noDiamond', '--blastn --min_id 0.45 --min_cov 50 -t 2 --min_ratio 0.25 -e 3,3 --gtable 11', None),
            ('map', '-f -m -O --blastn --diamond --min_id 0.4 --min_cov 50 --min_ratio 0.25 --merge_gap 600 --merge_diff 1.5 -t 1 -s 1 -e 0,3 --gtable 11', None),
            ('diamond_only_out', '--diamond --min_id 0.3 -t 1 -o OUT', None),
    ):
        d = fake_dir()
        qf, rf = os.path.join(d, 'q.fa'), os.path.join(d, 'r.fa')
        write_fasta(qf, genes, 60)
        write_fasta(rf, refs, 70)
        open(os.path.join(d, 'sam.0'), 'w').write(sam_text)
        open(os.path.join(d, 'bsn.0'), 'w').write(bsn_text)
        argv = ('-r %s -q %s ' % (rf, qf) + argv_tail).replace('OUT', os.path.join(d, 'out.tsv')).split()
        err = io.StringIO()
        with contextlib.redirect_stderr(err), contextlib.redirect_stdout(io.StringIO()):
            res = uberBlast.uberBlast(argv)
        rec = dict(name=name, argv=argv_tail)
        if isinstance(res, tuple):
            rec['rows'], rec['overlap'] = res
        else:
            rec['rows'] = res
        if os.path.exists(os.path.join(d, 'out.tsv')):
            rec['tsv'] = open(os.path.join(d, 'out.tsv')).read()
        out.append(rec)
    # empty results (uberBlast.py:356-359)
    d = fake_dir()
    qf, rf = os.path.join(d, 'q.fa'), os.path.join(d, 'r.fa')
    write_fasta(qf, genes); write_fasta(rf, refs)
    with contextlib.redirect_stderr(io.StringIO()), contextlib.redirect_stdout(io.StringIO()):
        e1 = uberBlast.uberBlast(('-r %s -q %s --blastn -t 1' % (rf, qf)).split())
        e2 = uberBlast.uberBlast(('-r %s -q %s --blastn -O -t 1' % (rf, qf)).split())
    out.append(dict(name='empty', shape=list(e1.shape), shape_O=[list(e2[0].shape), list(e2[1].shape)]))
    dump('g08_run.json', dict(query=genes, ref=refs, sam=sam_text, outfmt6=bsn_text, cases=out))


# ----------------------------------------------------------------------------- G9 getClust / iterClust
def g09():
    rng = np.random.default_rng(99)
    names = [str(x) for x in (5, 12, 7, 30, 31, 2, 18, 40, 41, 42, 9, 100)]
    seqs = {n: rand_cds(rng, int(rng.integers(20, 40))) for n in names}
    cases = []
    scripts = [
        # round tables: lines "rep\tmember"
        [[('5', '5'), ('5', '12'), ('7', '7'), ('30', '30'), ('30', '31'), ('2', '2'), ('18', '18'), ('18', '40'), ('41', '41'), ('42', '42'), ('9', '9'), ('100', '100')],
         [('7', '5'), ('7', '7'), ('30', '30'), ('2', '2'), ('2', '18'), ('41', '41'), ('42', '42'), ('9', '9'), ('100', '100')],
         [('7', '7'), ('30', '30'), ('2', '2'), ('41', '41'), ('41', '42'), ('9', '9'), ('100', '100')]],
        # stops after 2 rounds (no shrink)
        [[('12', '5'), ('12', '12'), ('7', '7'), ('31', '30'), ('31', '31'), ('2', '2'), ('18', '18'), ('40', '40'), ('41', '41'), ('42', '42'), ('9', '9'), ('100', '100')],
         [('5', '5'), ('7', '7'), ('30', '30'), ('2', '2'), ('18', '18'), ('40', '40'), ('41', '41'), ('42', '42'), ('9', '9'), ('100', '100')]],
    ]
    for si, script in enumerate(scripts):
        d = fake_dir()
        for i, tab in enumerate(script):
            with open(os.path.join(d, 'clust.tab.%d' % i), 'w') as f:
                for r, m in tab:
                    f.write('%s\t%s\n' % (r, m))
        gf = os.path.join(d, 'genes.fa')
        write_fasta(gf, seqs, 50)
        prefix = os.path.join(d, 'out')
        ex, tb = clust.getClust(prefix, gf, dict(identity=0.9, coverage=0.9, n_thread=2, translate=False))
        n_rounds = int(open(os.path.join(d, 'mm.count')).read())
        inputs = [open(os.path.join(d, 'mm.seen.%d' % i)).read() for i in range(n_rounds)]
        cases.append(dict(script=script, genes_fasta=open(gf).read(), exemplar=open(ex).read(), tab=open(tb).read(),
                          n_rounds=n_rounds, round_inputs=inputs))
    # translate=True branch (clust.py:38-46, 95-100)
    d = fake_dir()
    with open(os.path.join(d, 'clust.tab.0'), 'w') as f:
        for n in names:
            f.write('%s\t%s\n' % (names[0] if n in names[:3] else n, n))
    with open(os.path.join(d, 'clust.tab.1'), 'w') as f:
        for n in names[3:] + names[:1]:
            f.write('%s\t%s\n' % (n, n))
    gf = os.path.join(d, 'genes.fa')
    write_fasta(gf, seqs, 50)
    ex, tb = clust.getClust(os.path.join(d, 'out'), gf, dict(identity=0.9, coverage=0.9, n_thread=2, translate=True))
    cases.append(dict(translate=True, genes_fasta=open(gf).read(), exemplar=open(ex).read(), tab=open(tb).read(),
                      round_inputs=[open(os.path.join(d, 'mm.seen.0')).read()]))

    # iterClust: every getClust call consumes up to 3 scripted tables; use identity tables (no merging)
    # except for two steps, so the header-line loss (PEPPAN.py:1786) is visible
    d = fake_dir()
    gf = os.path.join(d, 'genes.fa')
    write_fasta(gf, seqs, 50)
    cur = list(names)
    k = 0
    plan = {0: [('5', '12')], 3: [('7', '5'), ('30', '31')], 10: [('2', '18')]}
    for step in range(11):
        merges = plan.get(step, [])
        gone = {m for r, m in merges}
        with open(os.path.join(d, 'clust.tab.%d' % k), 'w') as f:
            for n in cur:
                rep = [r for r, m in merges if m == n]
                f.write('%s\t%s\n' % (rep[0] if rep else n, n))
        k += 1
        cur = [n for n in cur if n not in gone]
        with open(os.path.join(d, 'clust.tab.%d' % k), 'w') as f:   # 2nd inner round: no change -> loop stops
            for n in cur:
                f.write('%s\t%s\n' % (n, n))
        k += 1
    groups = [[5, 900, 10000]]
    with contextlib.redirect_stderr(io.StringIO()):
        g = PEP.iterClust(os.path.join(d, 'it'), gf, groups, dict(identity=0.9, coverage=0.8, n_thread=2, translate=False))
    cases.append(dict(iterClust=True, genes_fasta=open(gf).read(), plan={str(k): v for k, v in plan.items()},
                      exemplar=open(g).read(), clust_npy=np.load(os.path.join(d, 'it.clust.npy')), final_tab=open(os.path.join(d, 'it.clust.tab')).read()))
    dump('g09_clust.json', dict(cases=cases))


# ----------------------------------------------------------------------------- G10/G11/G12
def pairs_table(rng):
    """self all-vs-all table shaped like uberBlast's output (sorted, CIGAR strings, str names)"""
    genes = {}
    rows = []
    ids = [3, 10, 11, 25, 26, 27, 40, 41, 55, 56, 57, 58, 70, 71, 80]
    for g in ids:
        genes[g] = int(rng.integers(40, 400)) * 3
    rid = 0

    def add(q, r, iden, qs, qe, ss, se, cigar=None):
        nonlocal rid
        ql, sl = genes[q], genes[r]
        if cigar is None:
            cigar = '%dM' % (qe - qs + 1)
        rows.append([str(q), str(r), iden, qe - qs + 1, 0, 0, qs, qe, ss, se, 0.0, float(int((qe - qs + 1) * (4 * iden - 1))), ql, sl, cigar, rid])
        rid += 1
    for g in ids:
        add(g, g, 1.0, 1, genes[g], 1, genes[g])
    # near identical, in frame, full length -> absorbed (direction decided by length + priority)
    genes[11] = genes[10]
    add(10, 11, 0.97, 1, genes[10], 1, genes[11]); add(11, 10, 0.97, 1, genes[11], 1, genes[10])
    genes[26] = genes[25] - 30
    add(25, 26, 0.95, 1, genes[25] - 30, 1, genes[26]); add(26, 25, 0.95, 1, genes[26], 1, genes[25] - 30)
    # reverse strand near identical -> conflict -2
    genes[41] = genes[40]
    add(40, 41, 0.99, 1, genes[40], genes[41], 1); add(41, 40, 0.99, 1, genes[41], genes[40], 1)
    # frame shifted
    add(27, 3, 0.93, 2, genes[27] - 2, 3, min(genes[3], genes[27] - 1))
    # orthologs below clust identity with indel CIGARs, several HSPs
    L = min(genes[55], genes[56])
    add(55, 56, 0.82, 1, 150, 1, 153, '60M3D90M'); add(55, 56, 0.78, 160, L - 3, 163, L, None)
    add(56, 55, 0.82, 1, 153, 1, 150, '60M3I90M')
    add(57, 58, 0.61, 4, 120, 4, 120); add(58, 57, 0.61, 4, 120, 4, 120)
    add(70, 71, 0.52, 10, 60, 10, 60)
    # repetitive pair: >= 50 rows
    for k in range(52):
        add(80, 3, 0.7, 1 + k, 60 + k, 1 + k, 60 + k)
    tab = pd.DataFrame(rows).sort_values([0, 1, 11]).values
    return genes, tab


def g10_g11_g12():
    rng = np.random.default_rng(1010)
    genes, tab = pairs_table(rng)
    d = fake_dir()
    cl = os.path.join(d, 'p.clust.exemplar')
    seqs = {g: ''.join('ACGT'[i] for i in rng.integers(0, 4, L)) for g, L in genes.items()}
    with open(cl, 'w') as f:
        for g, s in seqs.items():
            f.write('>%d\n%s\n' % (g, s))
    clu0 = np.array([[3, 900, 10000], [10, 901, 9950]], dtype=int)
    np.save(os.path.join(d, 'p.clust.npy'), clu0)
    prio = {g: [int(rng.integers(0, 3)), -L, int(rng.integers(0, 1 << 30))] for g, L in genes.items()}
    params = dict(noDiamond=False, match_identity=0.5, match_frag_len=50, n_thread=2, match_frag_prop=0.25, gtable=11,
                  clust_identity=0.9, clust_match_prop=0.8, incompleteCDS='', match_len=250., match_len1=100., match_len2=400.,
                  match_prop=0.5, match_prop1=0.8, match_prop2=0.4, clust=cl)
    calls = []

    def fake_uber(argv, pool=None):
        calls.append(argv)
        return copy.deepcopy(tab)
    PEP.uberBlast = fake_uber
    PEP.pool = None
    PEP.params = params
    res = PEP.get_similar_pairs(cl, prio, params)
    out = dict(table=tab, genes=genes, exemplar_in=''.join('>%d\n%s\n' % (g, s) for g, s in seqs.items()), clust_npy_in=clu0,
               priorities=prio, params={k: v for k, v in params.items() if k != 'clust'}, uber_argv=calls[0],
               pairs=res, exemplar_out=open(cl).read(), clust_npy_out=np.load(os.path.join(d, 'p.clust.npy'), allow_pickle=True))
    # incompleteCDS='sife' variant
    with open(cl, 'w') as f:
        for g, s in seqs.items():
            f.write('>%d\n%s\n' % (g, s))
    np.save(os.path.join(d, 'p.clust.npy'), clu0)
    params2 = dict(params, incompleteCDS='sife', clust_identity=0.99, match_identity=0.6)
    PEP.params = params2
    res2 = PEP.get_similar_pairs(cl, prio, params2)
    out['variant_sife'] = dict(params={k: v for k, v in params2.items() if k != 'clust'}, pairs=res2, exemplar_out=open(cl).read(),
                               clust_npy_out=np.load(os.path.join(d, 'p.clust.npy'), allow_pickle=True))
    dump('g10_pairs.json', out)

    # G11 get_gene_group
    cases = []
    for seed in (1, 2, 3):
        r = np.random.default_rng(seed)
        n = 60
        clu = np.array([[int(r.integers(0, n)), int(r.integers(0, n)), int(r.integers(9000, 10001))] for _ in range(40)], dtype=int)
        bsn = np.array([[int(a), int(b), int(v)] for a, b, v in zip(r.integers(0, n, 30), r.integers(0, n, 30), r.choice([-2, 0, 7000, 8123, 9500], 30))], dtype=int)
        np.save(os.path.join(d, 'g%d.clust.npy' % seed), clu)
        np.save(os.path.join(d, 'g%d.self_bsn.npy' % seed), bsn)
        grp = PEP.get_gene_group(os.path.join(d, 'g%d.clust.exemplar' % seed), os.path.join(d, 'g%d.self_bsn.npy' % seed))
        cases.append(dict(clu=clu, bsn=bsn, groups=[[int(k), [int(x) for x in v]] for k, v in grp.items()]))
    dump('g11_groups.json', dict(cases=cases))

    # G12 writeGenes
    r = np.random.default_rng(12)
    base = [rand_cds(r, int(c)) for c in (30, 30, 31, 30, 45)]
    genes12, prio12 = {}, {}
    order = [0, 1, 0, 2, 1, 3, 0, 4, 2, 0, 3]
    import hashlib
    for i, b in enumerate(order):
        s = base[b]
        genes12[i] = ['f', '', 0, 0, '+', int(hashlib.sha1(s.encode()).hexdigest(), 16), s]
        prio12[i] = [i % 2, -len(s), genes12[i][5]]
    genes12[99] = ['f', '', 0, 0, '+', 5, '']
    prio12[99] = [0, 0, 5]
    prio12[1000] = [0, 0, 0]   # a genome entry, not a gene
    fn, groups = PEP.writeGenes(os.path.join(d, 'w.genes'), genes12, prio12)
    dump('g12_writegenes.json', dict(genes={k: [v[5], v[6]] for k, v in genes12.items()}, priority=prio12,
                                     fasta=open(fn).read(), groups=groups))


def g13():
    """readers (configure.py:118-150, clust.py:7-18): multi-line FASTA with comments / lower case / descriptions, FASTQ, gz"""
    import gzip
    d = fake_dir()
    fasta = '>g1 first gene\nACGTacgtNN\nacgt\n#comment line\n>g2\n\nTTGA CC\n>g3 desc\nA\n'
    fastq = '@r1 some\nACGTN\n+\nIIII!\n@r2\nacgtacgt\n+r2\nABCDEFGH\n'
    out = {}
    for name, text in (('x.fa', fasta), ('x.fq', fastq)):
        p_ = os.path.join(d, name)
        open(p_, 'w').write(text)
        with gzip.open(p_ + '.gz', 'wt') as f:
            f.write(text)
        for q in (p_, p_ + '.gz'):
            seq, qual = configure.readFastq(q)
            out[os.path.basename(q)] = dict(readFastq=[seq, qual])
        if name.endswith('.fa'):
            out[name]['readFasta'] = configure.readFasta(p_)
            out[name]['readFasta_headOnly'] = configure.readFasta(p_, headOnly=True)
            out[name]['clust_readFasta'] = clust.readFasta(p_)
    dump('g13_readers.json', dict(fasta=fasta, fastq=fastq, out=out))


# ----------------------------------------------------------------------------- G14/G15 genome mapping callers
def nt_runs(ops):
    return [[3 * n, o] for n, o in rle(ops)]


def walk_identity(q, r, runs):
    i = j = mat = mis = 0
    for n, o in runs:
        if o == 'M':
            for k in range(n):
                if q[i + k] == r[j + k]:
                    mat += 1
                else:
                    mis += 1
            i += n; j += n
        elif o == 'I':
            i += n
        else:
            j += n
    gaps = sum(n for n, o in runs if o != 'M')
    return mat, mis, gaps


def map_world(rng, gid):
    """one genome (2 contigs, integer names) + the 17-column table uberBlast -f -m -O would hand to iter_map_bsn"""
    gene_ids = [5, 12, 13, 20, 31, 44, 45, 60, 61]
    genes = MAP_GENES
    contigs = {}
    rows = []
    old = {}

    def hit(g, cid, pos, allele, runs, strand, qs=1, qe=None, extra_iden=0.0):
        q = genes[g]
        qe = len(q) if qe is None else qe
        mat, mis, gaps = walk_identity(q[qs - 1:qe], allele, runs)
        alen = sum(n for n, o in runs)
        iden = round(mat / float(mat + mis + gaps), 3) - extra_iden
        score = float(3 * mat - mis - 5 * sum(1 for n, o in runs if o != 'M') - gaps)
        rl = sum(n for n, o in runs if o != 'I')
        ss, se = (pos + 1, pos + rl) if strand > 0 else (pos + rl, pos + 1)
        rows.append([str(g), str(cid), iden, alen, mis, sum(1 for n, o in runs if o != 'M'), qs, qe, ss, se, 0.0, score, len(q), None,
                     [list(r) for r in runs], len(rows)])

    for ci in range(2):
        cid = 1000 + 10 * gid + ci
        parts, pos = [], 0

        def spacer(n):
            nonlocal pos
            t = ''.join('ACGT'[i] for i in rng.integers(0, 4, n))
            parts.append(t); pos += n

        def place(g, sub, indel, strand, record_old=None, frame_shift=False, stop_at=None):
            nonlocal pos
            allele, ops = mutate_cds(rng, genes[g], sub, indel)
            runs = nt_runs(ops)
            if stop_at is not None:
                allele = allele[:stop_at * 3] + 'TAG' + allele[stop_at * 3 + 3:]
            if frame_shift:                               # drop one base inside the first M run -> 1I in the middle
                k = min(runs[0][0] - 10, 100)
                allele = allele[:k] + allele[k + 1:]
                runs = [[k, 'M'], [1, 'I'], [runs[0][0] - k - 1, 'M']] + runs[1:]
            start = pos
            parts.append(allele if strand > 0 else revcomp(allele)); pos += len(allele)
            hit(g, cid, start, allele, runs, strand)
            if record_old is not None:
                old.setdefault(str(cid), []).append([g, start + 1 + record_old, start + len(allele) + record_old, '+' if strand > 0 else '-', int(rng.integers(1, 1 << 40))])
            return start, allele, runs

        spacer(int(rng.integers(50, 300)))
        if ci == 0:
            st, al, ru = place(12, 0.02, 0.0, 1, record_old=0)
            # paralog 13 hits the same locus with a weaker alignment (same length genes) -> overlap pair
            a13, o13 = genes[13], None
            hit(13, cid, st, al[:len(genes[13])] if len(al) >= len(genes[13]) else al, [[min(len(al), len(genes[13])), 'M']], 1,
                qe=min(len(al), len(genes[13])))
            spacer(int(rng.integers(50, 300)))
            place(5, 0.05, 0.0, -1, record_old=0)
            spacer(int(rng.integers(50, 300)))
            place(20, 0.03, 0.0, 1, record_old=1, frame_shift=True)           # old prediction in another frame
            spacer(int(rng.integers(50, 300)))
            place(31, 0.01, 0.0, -1, stop_at=40)                               # premature stop inside the allele
            spacer(int(rng.integers(50, 300)))
            # gene 44 split into two collinear fragments 150 nt apart (linearMerge chains them); 45 competes on the first
            q = genes[44]
            h = (len(q) // 6) * 3
            a1, o1 = mutate_cds(rng, q[:h], 0.03, 0.0)
            a2, o2 = mutate_cds(rng, q[h:], 0.03, 0.0)
            s1 = pos; parts.append(a1); pos += len(a1)
            hit(44, cid, s1, a1, nt_runs(o1), 1, qs=1, qe=h)
            hit(45, cid, s1, a1[:min(len(a1), len(genes[45]))], [[min(len(a1), len(genes[45])), 'M']], 1, qs=1, qe=min(len(a1), len(genes[45])))
            spacer(150)
            s2 = pos; parts.append(a2); pos += len(a2)
            hit(44, cid, s2, a2, nt_runs(o2), 1, qs=h + 1, qe=len(q))
            old.setdefault(str(cid), []).append([44, s1 + 1, s2 + len(a2), '+', 77])
        else:
            st, al, ru = place(60, 0.04, 0.0, -1, record_old=0)
            hit(62, cid, st, al, [[len(al), 'M']], -1)                        # unlisted paralog pair (60, 62) -> overlap score 2
            spacer(int(rng.integers(50, 300)))
            # short partial hit (fails every match_len / match_prop rule -> identity set to -1)
            q = genes[61]
            a, o = mutate_cds(rng, q[:60], 0.02, 0.0)
            s0 = pos; parts.append(a); pos += len(a)
            hit(61, cid, s0, a, nt_runs(o), 1, qs=1, qe=60)
            spacer(int(rng.integers(50, 300)))
            place(5, 0.10, 0.0, 1)
            spacer(int(rng.integers(50, 300)))
            # reverse-strand fragments of gene 12 that overlap on the query (tests the max_sc overlap trimming)
            q = genes[12]
            h = (len(q) // 6) * 3
            a1, o1 = mutate_cds(rng, q[:h + 30], 0.02, 0.0)
            a2, o2 = mutate_cds(rng, q[h:], 0.06, 0.0)
            s2 = pos; parts.append(revcomp(a2)); pos += len(a2)
            hit(12, cid, s2, a2, nt_runs(o2), -1, qs=h + 1, qe=len(q))
            spacer(90)
            s1 = pos; parts.append(revcomp(a1)); pos += len(a1)
            hit(12, cid, s1, a1, nt_runs(o1), -1, qs=1, qe=h + 30)
        spacer(int(rng.integers(50, 300)))
        contigs[cid] = ''.join(parts)
    for r in rows:
        r[13] = len(contigs[int(r[1])])
    tab = np.empty([len(rows), 16], dtype=object)
    for i, r in enumerate(rows):
        for j, v in enumerate(r):
            tab[i, j] = v
    rb = uberBlast.RunBlast()
    tab = rb.linearMerge(tab, [True, 600., 1.5])
    rb.fixEnd(tab, 0, 3)
    ov = rb.returnOverlap(tab, [True, 300, 0.6])
    tab = pd.DataFrame(tab).sort_values([0, 1, 11]).values
    old = {c: sorted(v, key=lambda x: x[1]) for c, v in old.items()}
    return contigs, tab, ov, old


MAP_GENES = {}
MAP_PARAMS = dict(noDiamond=False, match_identity=0.65, match_frag_len=50, match_frag_prop=0.25, link_gap=600, link_diff=1.5, gtable=11,
                  match_len=250., match_len1=100., match_len2=400., match_prop=0.5, match_prop1=0.8, match_prop2=0.4)


class InOrderPool(object):
    def imap_unordered(self, fn, items):
        return map(fn, items)

    def close(self):
        pass

    def join(self):
        pass


def store_dump(fname):
    with PEP.MapBsn(fname) as c:
        return {k: c.get(k) for k in sorted(c.keys())}


def g14_g15():
    rng = np.random.default_rng(1414)
    for g in [5, 12, 20, 31, 44, 60, 61]:
        MAP_GENES[g] = rand_cds(rng, int(rng.integers(100, 300)))
    MAP_GENES[13] = mutate_cds(rng, MAP_GENES[12], 0.12, 0.0)[0]
    MAP_GENES[45] = mutate_cds(rng, MAP_GENES[44], 0.15, 0.0)[0]
    MAP_GENES[62] = mutate_cds(rng, MAP_GENES[60], 0.10, 0.0)[0]
    d = fake_dir()
    os.chdir(d)
    clust_fn = os.path.join(d, 'm.clust.exemplar')
    with open(clust_fn, 'w') as f:
        for g, s in MAP_GENES.items():
            f.write('>%d\n%s\n' % (g, s))
    self_bsn = np.array([[12, 13, 8800], [44, 45, -2], [5, 60, 0], [20, 31, 7000]], dtype=int)
    np.save(os.path.join(d, 'm.self_bsn.npy'), self_bsn)
    worlds = [map_world(rng, k) for k in range(3)]
    old_fn = os.path.join(d, 'm.old_prediction.npz')
    old_all = {}
    with PEP.MapBsn(old_fn, 'w') as op:
        for contigs, tab, ov, old in worlds:
            for c, v in old.items():
                op.save(c, np.array(v, dtype=object))
                old_all[c] = v
    tables = {}
    calls = []

    def fake_uber(argv, pool=None):
        calls.append(list(argv))
        gfile = argv[argv.index('-r') + 1]
        first = open(gfile).readline()[1:].strip()
        tab, ov = tables[int(first)]
        return copy.deepcopy(tab), ov.copy()
    PEP.uberBlast = fake_uber
    PEP.params = dict(MAP_PARAMS)
    # --- G14: iter_map_bsn + compare_prediction, one genome at a time
    cases = []
    for k, (contigs, tab, ov, old) in enumerate(worlds):
        tables[min(contigs)] = (tab, ov)
        seq = [[c, s] for c, s in contigs.items()]
        cp = PEP.compare_prediction(np.array([list(r) for r in copy.deepcopy(tab)], dtype=object), old_fn)
        pref = PEP.iter_map_bsn((os.path.join(d, 'm'), clust_fn, k, 'taxon%d' % k, seq, os.path.join(d, 'm.self_bsn.npy'), old_fn, dict(MAP_PARAMS)))
        z = np.load(pref + '.bsn.npz', allow_pickle=True)
        cases.append(dict(contigs=contigs, table=tab, overlap=ov, compare_prediction=cp, bsn=z['bsn'], ovl=z['ovl'], argv=calls[-1][4:]))
        os.unlink(pref + '.bsn.npz')
    seqs = np.array([[0, 1, 2, 3, 4, 124], [31, 62, 93, 7, 100, 55]], dtype=np.uint8)
    dump('g14_mapbsn.json', dict(genes=MAP_GENES, self_bsn=self_bsn, params=MAP_PARAMS, old_prediction=old_all, cases=cases,
                                 decodeSeq_in=seqs, decodeSeq_out=PEP.decodeSeq(seqs)))
    # --- G15: get_map_bsn over the three genomes, in-order pool -> the four MapBsn stores
    genomes = {}
    for k, (contigs, tab, ov, old) in enumerate(worlds):
        for c, s in contigs.items():
            genomes[c] = [900 + k, s]
    PEP.pool = InOrderPool()
    out = {}
    for save_seq in (True, False):
        names = [os.path.join(d, 'mm%d.%s.npz' % (save_seq, x)) for x in ('tab', 'seq', 'mat', 'conflicts')]
        with PEP.MapBsn(names[0], 'w') as c0, PEP.MapBsn(names[1], 'w') as c1, PEP.MapBsn(names[2], 'w') as c2, PEP.MapBsn(names[3], 'w') as c3:
            PEP.get_map_bsn(os.path.join(d, 'm'), clust_fn, genomes, os.path.join(d, 'm.self_bsn.npy'), old_fn, c0, c1, c2, c3, save_seq)
        out['saveSeq_%d' % save_seq] = {x: store_dump(n) for x, n in zip(('tab', 'seq', 'mat', 'conflicts'), names)}
    dump('g15_getmapbsn.json', dict(genomes={str(c): v[0] for c, v in genomes.items()}, stores=out))
    os.chdir(HERE)


# ----------------------------------------------------------------------------- G16 real genes (the reference's examples/)
def g16_real():
    """the first ~450 kb of the main chromosome of the four example genomes through the reference's own front end
    (iter_readGFF, checkPseu, writeGenes; PEPPAN.py:117-182, 989-1010, 1023-1039): real gene sequences, their sha1 codes,
    priorities, the reference's duplicate groups and unique-gene order - data for the K13 / search / clustering tests"""
    import gzip, glob
    PEP.params = dict(min_cds=120, incompleteCDS='')
    genes, priority, contig_piece = {}, {}, None
    gid = 0
    for rank, fn in enumerate(sorted(glob.glob(os.path.join(REF, 'examples', '*.gff.gz')))):
        seq, cds = PEP.iter_readGFF((fn, 'CDS', 11))
        main = max(seq, key=lambda n: len(seq[n][1]))
        if contig_piece is None:
            contig_piece = seq[main][1][:300000]
        picked = sorted((c[2], n) for n, c in cds.items() if c[1] == main and c[3] < 450000 and len(c[6]))
        for start, n in picked:
            c = cds[n]
            genes[gid] = [os.path.basename(fn), c[1], c[2], c[3], c[4], c[5], c[6]]
            priority[gid] = [rank // 2, -len(c[6]), c[5]]          # two priority classes of two genomes each
            gid += 1
    d = fake_dir()
    fn, groups = PEP.writeGenes(os.path.join(d, 'real.genes'), genes, priority)
    unique_order = [int(l[1:]) for l in open(fn) if l.startswith('>')]
    with gzip.open(os.path.join(HERE, 'g16_real_genes.fa.gz'), 'wt') as f:
        for g, v in genes.items():
            f.write('>%d\n%s\n' % (g, v[6]))
    with gzip.open(os.path.join(HERE, 'g16_real_contig.fa.gz'), 'wt') as f:
        f.write('>900001\n%s\n' % contig_piece)
    dump('g16_real.json', dict(n_genes=len(genes), hash={g: str(v[5]) for g, v in genes.items()}, priority={g: [p[0], p[1], str(p[2])] for g, p in priority.items()},
                               meta={g: v[:5] for g, v in genes.items()}, groups=groups, unique_order=unique_order))
    print('g16: %d gene instances, %d unique, %d duplicate pairs' % (len(genes), len(unique_order), len(groups)))


# ----------------------------------------------------------------------------- G17 the whole example data set (BASELINE configs[0])
def g17_examples():
    """every CDS of the reference's four example genomes through the steps PEPPAN.ortho() takes before the first external call
    (PEPPAN.py:1844-1849, 1880): readGFF (iter_readGFF per file; its worker pool replaced by a plain map), encodeNames, load_priority
    (default: no priority list), writeGenes.  Recorded: the unique genes exactly as the reference wrote <prefix>.genes (integer names),
    the priority of every one of them, and - for the duplicate collapse - (name, length, sha1 code, priority) of every gene instance with
    the reference's duplicate groups."""
    import gzip, glob

    class _Serial(object):
        def imap_unordered(self, fn, jobs):
            return map(fn, jobs)
    PEP.pool = _Serial()
    PEP.params = dict(min_cds=120, incompleteCDS='')
    files = sorted(glob.glob(os.path.join(REF, 'examples', '*.gff.gz')))
    genomes, genes = PEP.readGFF(files, 'CDS', 11)
    d = fake_dir()
    n_cds = len(genes)
    genomes, genes, encodes = PEP.encodeNames(genomes, genes, '', os.path.join(d, 'ex.encode.csv'), False)
    priorities = PEP.load_priority('', genes, encodes)
    n_with_seq = sum(1 for g in genes.values() if len(g[6]))
    fn, groups = PEP.writeGenes(os.path.join(d, 'ex.genes'), genes, priorities)
    unique = [int(l[1:]) for l in open(fn) if l.startswith('>')]
    with open(fn) as fin, gzip.open(os.path.join(HERE, 'g17_examples_genes.fa.gz'), 'wt', compresslevel=9) as f:
        f.write(fin.read())
    inst = sorted(n for n, g in genes.items() if len(g[6]))
    dump('g17_examples.json', dict(n_cds=n_cds, n_instances=n_with_seq, n_unique=len(unique),
                                   priority={g: [priorities[g][0], priorities[g][1], str(priorities[g][2])] for g in unique},
                                   instances=[[g, len(genes[g][6]), '%040x' % genes[g][5], priorities[g][0]] for g in inst],
                                   groups=groups, nt_total=sum(len(genes[g][6]) for g in unique)))
    print('g17: %d CDS, %d instances with a sequence, %d unique genes, %d duplicate pairs' % (n_cds, n_with_seq, len(unique), len(groups)))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] in ('g17', 'g18'):
        {'g17': g17_examples, 'g18': g18}[sys.argv[1]]()
        shutil.rmtree(SHIM, ignore_errors=True)
        sys.exit(0)
    g13()
    g01()
    genes, refs, rel = g02()
    sam_text = g03(genes, refs, rel)
    bsn_text = g04(genes, refs, rel)
    g05_g06(genes, refs, sam_text, bsn_text)
    g07()
    g18()
    g08(genes, refs, sam_text, bsn_text)
    g09()
    g10_g11_g12()
    g14_g15()
    g16_real()
    g17_examples()
    shutil.rmtree(SHIM, ignore_errors=True)
