"""Bit-exactness of the whole search against the CPU oracle at a size of your choice (the -m gpu suite stops at 50 000 genes):
python3 tools/parity_at_size.py <genes> [gene_len, 0 = log-normal lengths]     - every field of every hit, the CIGAR arena, the statistics"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from peppan_amd import _native as N, synth
from oracle import oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
gene_len = int(sys.argv[2]) if len(sys.argv) > 2 else 1002
names, seqs = synth.make_genes(n, gene_len, seed=355)
order = sorted(range(len(names)), key=lambda i: names[i])
nts = [seqs[i] for i in order]
ctx = N.Context(0)
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
O.lib().oracle_set_threads(0)
t0 = time.perf_counter()
oh, oc, ost = O.search(q_aa, t_aa, O.default_params(45., 25., 10, 5))
t_cpu = time.perf_counter() - t0
bad = [f for f in ('q', 't', 'q_start', 'q_end', 't_start', 't_end', 'score', 'nm', 'n_ident', 'aln_len', 'cigar_runs', 'bin', 'cigar_off', 'cells')
       if len(gh) != len(oh) or not np.array_equal(gh[f], oh[f])]
bad += ['cigar arena'] if not np.array_equal(gc, oc) else []
bad += ['stat ' + k for k in ('candidates', 'pairs', 'cells', 'tracebacks') if st[k] != ost[k]]
print('%d genes x %s nt all-vs-all: GPU search %.1f ms (K1 inside: no), oracle %.1f s on the CPUs the box grants; %d candidates (%d identical pairs scored by comparison), '
      '%d traced pairs (%d one ungapped run), %d hits, %d CIGAR runs: %s'
      % (n, gene_len or 'log-normal', t_gpu * 1e3, t_cpu, st['candidates'], st['candidates_settled'], st['tracebacks'], st['tracebacks_gapless'], len(gh), len(gc),
         'IDENTICAL to the oracle in every field' if not bad else 'DIFFERENT: ' + ', '.join(bad)))
sys.exit(1 if bad else 0)
