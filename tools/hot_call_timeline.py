"""The reference's hot call through the drop-in, as a TIMELINE of one call: python tools/hot_call_timeline.py [n_genes] [reps]
Every library call and every step of the host chain is recorded with its thread, start and end (perf_counter); the call with the median wall
time is printed, intervals in start order, the side thread (the nucleotide tool) indented.  Then the same for get_similar_pairs (the search
in front of it re-reads the exemplar file the step before rewrote).  profiles/r06_hot_call_timeline.txt is this tool's output."""
import contextlib, io, os, shutil, sys, tempfile, threading, time
sys.path.insert(0, '.')
os.environ.setdefault('PEPPAN_LOG', '0')
import numpy as np
from peppan_amd import synth, uberBlast as UB, _native as N, pipeline as PL, configure as CF, hittable as HT, mapfilters as MF
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 15
names, seqs = synth.make_genes(n, 1002, seed=355)
order = sorted(range(len(names)), key=lambda i: names[i])
events = []
main = threading.get_ident()


def timed(obj, name, label=None):
    f = getattr(obj, name)
    lab = label or name

    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            events.append((threading.get_ident() != main, lab, t, time.perf_counter()))
    try:
        setattr(obj, name, g)
    except (AttributeError, TypeError):
        pass


def show(title, calls):
    calls.sort(key=lambda c: c[0])
    wall, t0, ev = calls[len(calls) // 2]
    print('%s: median of %d calls %.2f ms (fastest %.2f)' % (title, len(calls), wall, calls[0][0]))
    for side, lab, a, b in sorted(ev, key=lambda e: (e[2], -e[3])):
        print('  %s%8.2f .. %7.2f  %6.2f ms  %s' % ('        | ' if side else '', (a - t0) * 1e3, (b - t0) * 1e3, (b - a) * 1e3, lab))


with tempfile.TemporaryDirectory() as tmp:
    fa = os.path.join(tmp, 'exemplar.fa')
    with open(fa, 'w') as f:
        for i in order:
            f.write('>%s\n%s\n' % (names[i], seqs[i].decode()))
    argv = ('-r %s -q %s --blastn --diamond -s 1 --min_id 0.45 --min_cov 50 -t 1 --min_ratio 0.25 -e 3,3 -p --gtable 11' % (fa, fa)).split()
    with contextlib.redirect_stderr(io.StringIO()):
        UB.uberBlast(argv); UB.uberBlast(argv)
        for ctx, tag in ((UB.get_context(), 'ctx'), (UB.get_nucl_context(), 'nctx')):
            for nm in ('search', 'use_nt_as_residues', 'translate', 'rescore_nt', 'target_meta', 'query_meta', 'set_query_nt', 'set_ref_nt', 'pair_support', 'overlaps'):
                timed(ctx, nm, tag + '.' + nm)
        for nm in ('table_from_hits', 'cols_gather', 'cols_order', 'cols_fix_end', 'similar_scan', 'similar_resolve', 'fasta_keep', 'fasta_records', '_pack'):
            timed(N, nm, 'N.' + nm)
        for nm in ('to_rows', 'take', 'fix_end', 'final_order', 'concat'):
            timed(HT.HitTable, nm, 'HitTable.' + nm)
        for nm in ('_rescore_table', '_runBlast_table', '_runDiamond_table', '_post', '_run_tools', '_ensure_nt', '_load', 'run'):
            timed(UB.RunBlast, nm, 'RunBlast.' + nm)
        for nm in ('_read_cached', '_prepare_side', 'hits_to_table', 'blast_hits_to_table', '_parser', '_run_arguments'):
            timed(UB, nm, 'UB.' + nm)
        for nm in ('_classify_rows', '_self_search', '_drop_dead_exemplars'):
            timed(PL, nm, 'PL.' + nm)
        timed(CF, 'readFastq', 'configure.readFastq')
        UB.readFastq = CF.readFastq
        calls = []
        for _ in range(reps):
            del events[:]
            t = time.perf_counter()
            tab = UB.uberBlast(argv)
            calls.append(((time.perf_counter() - t) * 1e3, t, list(events)))
            del tab                                   # (the object rows of a call are released outside the next call's time)
        show('uberBlast --blastn --diamond -s 1, %d genes against themselves, object rows' % n, calls)
        calls = []
        for _ in range(reps):
            del events[:]
            t = time.perf_counter()
            tab = UB.uberBlast(argv, as_table=True)
            calls.append(((time.perf_counter() - t) * 1e3, t, list(events)))
        show('the same call with as_table=True (numeric table, what get_similar_pairs takes)', calls)
        # get_similar_pairs
        prio = {int(names[i]): [0, -len(seqs[i]), int(names[i])] for i in order}
        params_gs = dict(noDiamond=False, match_identity=0.5, match_frag_len=50, n_thread=1, match_frag_prop=0.25, gtable=11, clust_identity=0.9, clust_match_prop=0.8,
                         incompleteCDS='', match_len=250., match_len1=100., match_len2=400., match_prop=0.5, match_prop1=0.8, match_prop2=0.4)
        ex = os.path.join(tmp, 'p.clust.exemplar')
        np.save(os.path.join(tmp, 'p.clust.npy'), np.zeros((0, 3), dtype=int))
        calls = []
        for _ in range(max(5, reps // 2)):
            shutil.copy(fa, ex)
            np.save(os.path.join(tmp, 'p.clust.npy'), np.zeros((0, 3), dtype=int))
            del events[:]
            t = time.perf_counter()
            pairs = PL.get_similar_pairs(ex, prio, dict(params_gs, clust=ex))
            calls.append(((time.perf_counter() - t) * 1e3, t, list(events)))
        show('get_similar_pairs (exemplar file new for every call), %d pairs' % len(pairs), calls)
