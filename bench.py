#!/usr/bin/env python3
"""bench.py - one "step" = one full pass of the hot path over the 10k-gene all-vs-all workload:
K1 translate/pack -> K2-K4 seeds/candidates -> K5 banded Smith-Waterman -> K6 traceback -> K8 filters/top-k ->
hit table to the host -> (N>1: RCCL all-gather of the shard hit tables + top-k merge) -> K10 union-find.
Inputs (nucleotides) are resident in HBM before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--genes G] [--no-cpu-baseline]

N>1: one rank per GPU on an R x C grid of (query range, reference range) shards (peppan_amd/dist.py).  Either the launcher provides
RANK / WORLD_SIZE (torch.distributed.run), or - when RANK is unset - this script starts the N ranks itself and relays rank 0's line.
Strong scaling: the total workload is the named 10k x 10k configuration whatever N is."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

def _profile(name):
    """the newest tracked copy of a profile file: this round's (r06) when tools/profile_round.sh has been run on the final code, else round 5's"""
    for rnd in ('r06', 'r05'):
        path = os.path.join(ROOT, 'profiles', '%s_%s' % (rnd, name))
        if os.path.exists(path):
            return path
    return os.path.join(ROOT, 'profiles', 'r06_' + name)


PROFILE_COUNTERS = _profile('counters.json')                 # rocprofv3 PMC passes of the headline workload (tools/profile_round.sh)
PROFILE_COUNTERS_50K = _profile('counters_50k.json')         # the same passes over the 50 000-gene all-vs-all (the `workloads` block)
PROFILE_COUNTERS_BLASTN = _profile('counters_blastn.json')   # ... over the nucleotide tool on the headline's genes (tools/one_search.py 10000 blastn)
PROFILE_COUNTERS_MAP50K = _profile('counters_map50k.json')   # ... over one mapping step of the map_50k leg (tools/one_map_step.py)
PROFILE_VALU = _profile('valu_rate.txt')                     # tools/micro/valu_rate on the same GPU
PROFILE_VALU_MIX = _profile('valu_mix.json')                 # tools/valu_mix.py: the sweep loops' instruction mix from the compiler's assembly


REFERENCE_TOOLS = ('diamond', 'blastn', 'makeblastdb', 'mmseqs')


def probe_reference_tools():
    """which of the binaries the reference shells out to exist on this machine (its bundled copies are stripped,
    .MISSING_LARGE_BLOBS), and what CPU this is"""
    import shutil
    found = {t: shutil.which(t) for t in REFERENCE_TOOLS}
    model = None
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                model = line.split(':', 1)[1].strip()
                break
    except OSError:
        pass
    return found, model


def reference_cpu_path(found, names, nts, min_id, min_ratio, threads):
    """The reference's own CPU path on the same genes, when its binaries are installed: the IDENTICAL command lines of uberBlast.py:531, 550
    (diamond makedb + 5 x blastp on the 5 round-robin splits of the translated reference), uberBlast.py:492, 294 (makeblastdb + one blastn per
    query shard, `threads` shards side by side) and clust.py:62-66 (mmseqs createdb / linclust / createtsv), timed with time.perf_counter.
    Inputs are the files the reference itself would write (qryAA / refAA.0-4 text: oracle.diamond_fasta, pinned to golden G2)."""
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    out = {}
    run = lambda cmd, **kw: subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, **kw)
    with tempfile.TemporaryDirectory() as d:
        genes = {n: s.decode() for n, s in zip(names, nts)}
        if found.get('diamond'):
            q_txt, r_txt = O.diamond_fasta(genes, genes, '7', 11)
            qry, ref, hit = os.path.join(d, 'qryAA'), os.path.join(d, 'refAA'), os.path.join(d, 'aaMatch')
            open(qry, 'w').write(q_txt)
            t0 = time.perf_counter()
            run('{0} makedb --db {1} --in {1}'.format(found['diamond'], qry).split())
            n_rec = 0
            for i in range(5):
                open('%s.%d' % (ref, i), 'w').write(r_txt[i])
                run('{diamond} blastp --no-self-hits --threads {n_thread} --db {refAA} --query {qryAA} --out {aaMatch} --id {min_id} --query-cover {min_ratio} --evalue 1 -k {nhits} --dbsize 5000000 --outfmt 101'.format(
                    diamond=found['diamond'], refAA='%s.%d' % (ref, i), qryAA=qry, aaMatch='%s.%d' % (hit, i), n_thread=threads, min_id=min_id, nhits=10, min_ratio=min_ratio).split())
                if os.path.exists('%s.%d' % (hit, i)):
                    n_rec += sum(1 for line in open('%s.%d' % (hit, i)) if not line.startswith('@'))
            out['diamond'] = dict(seconds=time.perf_counter() - t0, records=n_rec, version=subprocess.run([found['diamond'], 'version'], capture_output=True, text=True).stdout.strip())
        if found.get('blastn') and found.get('makeblastdb'):
            ref = os.path.join(d, 'refNA')
            with open(ref, 'w') as f:
                for n, s in genes.items():
                    f.write('>{0}\n{1}\n'.format(n, s))
            t0 = time.perf_counter()
            run('{0} -dbtype nucl -in {1} -out {1}'.format(found['makeblastdb'], ref).split())
            order = sorted(genes.items(), key=lambda kv: -len(kv[1]))
            shards = [os.path.join(d, 'qryNA.%d' % i) for i in range(min(len(order), threads))]
            for i, q in enumerate(shards):
                with open(q, 'w') as f:
                    for n, s in order[i::threads]:
                        f.write('>{0}\n{1}\n'.format(n, s))
            cmd = ('{blastn} -db {refDb} -query {qry} -word_size 17 -out {qry}.bsn -perc_identity {min_id} -outfmt "6 qseqid sseqid pident length mismatch gapopen qstart qend sstart send evalue score qlen slen qseq sseq" '
                   '-qcov_hsp_perc {min_ratio} -num_alignments 1000 -task blastn -evalue 1e-2 -dbsize 5000000 -reward 2 -penalty -3 -gapopen 6 -gapextend 2')
            with ThreadPoolExecutor(max_workers=threads) as pool:
                list(pool.map(lambda q: run(cmd.format(blastn=found['blastn'], refDb=ref, qry=q, min_id=min_id, min_ratio=min_ratio), shell=True), shards))
            out['blastn'] = dict(seconds=time.perf_counter() - t0, records=sum(sum(1 for _ in open(q + '.bsn')) for q in shards if os.path.exists(q + '.bsn')))
        if found.get('mmseqs'):
            fa, db, lc, tmp, tab = (os.path.join(d, x) for x in ('genes.fa', 'seq.db', 'seq.lc', 'tmp', 'clust.tab'))
            with open(fa, 'w') as f:
                for n, s in genes.items():
                    f.write('>{0}\n{1}\n'.format(n, s))
            os.makedirs(tmp)
            t0 = time.perf_counter()
            run('{0} createdb {2} {1} -v 0'.format(found['mmseqs'], db, fa).split())
            run('{0} linclust {1} {2} {3} --min-seq-id {4} -c {5} --threads {6} -v 0'.format(found['mmseqs'], db, lc, tmp, 0.9, 0.9, threads).split())
            run('{0} createtsv {1} {1} {2} {3}'.format(found['mmseqs'], db, lc, tab).split())
            out['mmseqs'] = dict(seconds=time.perf_counter() - t0, records=sum(1 for _ in open(tab)) if os.path.exists(tab) else 0)
    return out


def cpu_baseline(names, nts, n_sample, min_id, min_qcov):
    """The CPU figure beside the GPU line.  The reference's binaries are probed for first (and timed with the reference's own command lines when
    they exist: kind "reference"); on a machine without them - the GPU boxes of this pool - it is the CPU oracle (C port: seed phase parallel
    over targets, alignment phase over (query, target) pairs, OpenMP on every host core) on a bounded sample of the same workload: the first
    n_sample queries against the whole reference, with its phase split, plus the rate a VECTORISED Smith-Waterman reaches on the same CPU
    (oracle/full_sw.c: 32 targets per query side by side, all cores) - the port is deliberately plain scalar code and says nothing about what
    DIAMOND or a SIMD aligner would do.  Reported beside the GPU number, never the target."""
    from oracle import oracle as O
    from peppan_amd.configure import transeq          # numpy translation (pinned to the same golden vectors as the oracle's)
    found, cpu_model = probe_reference_tools()
    from peppan_amd.configure import effective_cpus
    visible = len(os.sched_getaffinity(0))
    cores = effective_cpus()                          # (the control group's CPU allowance counts: 256 threads on 16 CPUs' worth of time are 16 cores)
    have = sorted(t for t, path in found.items() if path)
    sys.stderr.write('bench.py: reference binaries on PATH: %s; CPU: %s, %d hardware threads visible, %d CPUs granted to this container\n' % (
        ', '.join('%s=%s' % (t, found[t]) for t in have) or 'none', cpu_model, visible, cores))
    reference = None
    if have:
        try:
            reference = reference_cpu_path(found, names, nts, min_id, min_qcov, cores)
        except Exception as e:                          # a broken installation must not cost the line
            reference = {'error': repr(e)}
    q_aa = []
    for n, frames in transeq([[i, s.decode()] for i, s in enumerate(nts[:n_sample])], frame='F', transl_table=11):
        q_aa.append(O.aa_codes(min((f[:-1].count('X'), k, f) for k, f in enumerate(frames))[2].replace('-', 'X')))
    t_aa = []
    for n, frames in transeq([[i, s.decode()] for i, s in enumerate(nts)], frame='7', transl_table=11):
        for aa in frames:
            t_aa += [O.aa_codes(c.replace('-', 'X')) for o, c in O.ref_chunks(aa)]
    O.lib().oracle_set_threads(cores)
    O.phase_seconds(reset=True)
    t0 = time.perf_counter()
    hits, cig, st = O.search(q_aa, t_aa, O.default_params(min_id, min_qcov, 10, 5))
    dt = time.perf_counter() - t0
    ph = O.phase_seconds()
    # a vectorised CPU Smith-Waterman on the same residues: every (query, target frame) pair of a slice of the workload, full matrix, score only
    vec = None
    try:
        from oracle import full_sw as F
        nq_v, nt_v = min(64, len(q_aa)), min(len(t_aa), 12000)
        t1 = time.perf_counter()
        F.score_matrix(q_aa[:nq_v], t_aa[:nt_v], O.default_params(min_id, min_qcov, 10, 5), threads=cores)
        dv = time.perf_counter() - t1
        cells_v = float(sum(len(x) for x in q_aa[:nq_v])) * float(sum(len(x) for x in t_aa[:nt_v]))
        vec = dict(cells_per_s=cells_v / dv, seconds=dv, sample='%d queries x %d target frames, full matrices, score only, %d threads (oracle/full_sw.c, 32 targets per query in SIMD lanes)' % (nq_v, nt_v, cores))
    except Exception as e:
        vec = {'error': repr(e)}
    line = dict(value=st['candidates'] / dt, unit='gene-pairs/s', cores=cores, hardware_threads_visible=visible, kind='port', cpu_model=cpu_model,
                reference_binaries={t: found[t] for t in REFERENCE_TOOLS}, reference_cpu_path=reference,
                sample='first %d of %d queries vs all %d genes x 6 frames; %.1f s on %d threads; %d candidates, %.3g SW cells (%.3g cells/s)'
                       % (n_sample, len(nts), len(nts), dt, cores, st['candidates'], st['cells'], st['cells'] / dt),
                seconds=dt, sw_cells_per_s=st['cells'] / dt,
                phase_s=dict(seed=ph['seed_s'], align=ph['align_s'], other=max(0., dt - ph['seed_s'] - ph['align_s']),
                             score_pass_share_of_align=ph['score_thread_s'] / max(1e-12, ph['score_thread_s'] + ph['trace_thread_s']),
                             traceback_share_of_align=ph['trace_thread_s'] / max(1e-12, ph['score_thread_s'] + ph['trace_thread_s'])),
                sw_cells_per_s_vectorised=vec.get('cells_per_s') if vec else None, vectorised_sw=vec,
                note='kind "port": the reference\'s binaries (diamond / blastn / mmseqs) are absent from this machine (probed: reference_binaries), so the figure is the '
                     'oracle C port - scalar banded code under OpenMP; sw_cells_per_s_vectorised is what a SIMD Smith-Waterman sustains on the same cores. With the '
                     'binaries on PATH the same run times the reference\'s own command lines (reference_cpu_path) and reports kind "reference"',
                _hits=(hits, cig, n_sample))
    if reference and reference.get('diamond', {}).get('seconds'):
        d = reference['diamond']
        line.update(kind='reference', value=float(len(nts)) * float(len(nts)) / d['seconds'], unit='gene pairs of the all-vs-all/s (diamond makedb + 5 x blastp, wall)', cores=cores,
                    sample='all %d genes, the reference\'s diamond command lines (uberBlast.py:531, 550), %d threads, %.1f s, %d SAM records' % (len(nts), cores, d['seconds'], d['records']))
    return line


def spawn_ranks(args, argv):
    """--gpus N without a launcher: this (GPU-free) parent starts one child per rank and relays rank 0's JSON line"""
    import socket
    import torch
    n = args.gpus
    share = os.environ.get('PEPPAN_BENCH_SHARE_GPU') == '1'
    have = torch.cuda.device_count()                   # counting devices does not initialise the GPU
    if have < n and not share:
        sys.stderr.write('bench.py: --gpus %d but %d device(s) visible (PEPPAN_BENCH_SHARE_GPU=1 runs all ranks on device 0 over gloo)\n' % (n, have))
        return 2
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = procs[0].communicate()[0].decode()
    rcs = [p.wait() for p in procs]
    sys.stdout.write(out0)
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(rcs) if c != 0]
    if bad:
        sys.stderr.write('bench.py: ranks failed: %r\n' % bad)
        return 1
    return 0


def map_workload(args, rank, world, local_rank, n_genomes, steps, warmup, with_stores=False, gene_set=None, presence=None, timing=None, search_log=None):
    """The genes -> genomes mapping (BASELINE configs "x 500 / x 2000 genomes"; PEPPAN.py:907-989) as a bench workload: every rank maps ITS OWN
    n_genomes synthetic genomes (10 000 exemplar genes, 2.2 Mb per genome) - batched GPU search for both tools, -f / -m / -O chain, K7, K12,
    build_bsn - per step.  Genomes are the independent unit of this path: they shard over the ranks with no data-path collective (weak
    scaling), which is what can scale on an 8-GPU node where the 3 ms all-vs-all step cannot."""
    import contextlib
    import io
    import tempfile
    from peppan_amd import mapbsn, synth, uberBlast as UB
    names, seqs = gene_set or synth.make_genes(args.genes, 0, seed=355)
    params = dict(noDiamond=False, match_identity=0.65, match_frag_len=50, match_frag_prop=0.25, link_gap=600, link_diff=1.5, gtable=11,
                  match_len=250., match_len1=100., match_len2=400., match_prop=0.5, match_prop1=0.8, match_prop2=0.4)
    tmp = tempfile.mkdtemp(prefix='pep_map_%d_' % rank)
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        with open('m.clust.exemplar', 'w') as f:
            for i, s in enumerate(seqs):
                f.write('>%d\n%s\n' % (i, s.decode()))
        worlds = synth.make_genomes(seqs, n_genomes, seed=355 + 1000 * rank, presence=presence)
        jobs, nt = [], 0
        with mapbsn.MapBsn('m.old_prediction.npz', 'w') as op:
            for g, (gname, contig, ann) in enumerate(worlds):
                jobs.append((g, 900000 + g, [[100000 + g, contig.decode()]]))
                nt += len(contig)
                op.save(100000 + g, np.array([[k, a, b, st, 1] for k, a, b, st in ann[::2]], dtype=object))
        og = np.array([[0, 1, 9000], [4, 5, -2]], dtype=int)
        np.save('m.self_bsn.npy', og)
        UB._CTX.clear()
        os.environ['PEPPAN_HIP_DEVICE'] = str(local_rank)
        if search_log is not None:
            # every search of the drop-in's context with all phase timers on, its statistics appended to the caller's list
            sctx = UB.get_context()
            sctx.set_timing(2)
            plain_search = sctx.search

            def recorded(*a, **k):
                r = plain_search(*a, **k)
                search_log.append(r[2])
                return r
            sctx.search = recorded
        ortho = mapbsn.OrthoRelation(og)
        genomes = {100000 + g: [900000 + g, contig.decode()] for g, (gname, contig, ann) in enumerate(worlds)}

        def step():
            groups = rows = 0
            for job, (tab, ovl) in zip(jobs, mapbsn._gpu_search('m', 'm.clust.exemplar', jobs, params, genomes_per_batch=n_genomes)):
                G = mapbsn.build_groups(tab, ovl, job[2], ortho, 'm.old_prediction.npz', params)
                groups += len(G)
                rows += len(tab)
            return groups, rows

        def step_with_stores():
            """the same genomes through get_map_bsn with the reference's four stores written (PEPPAN.py:907-989; .seq included)"""
            with mapbsn.MapBsn('t.npz', 'w') as c0, mapbsn.MapBsn('s.npz', 'w') as c1, mapbsn.MapBsn('a.npz', 'w') as c2, mapbsn.MapBsn('c.npz', 'w') as c3:
                mapbsn.get_map_bsn('m', 'm.clust.exemplar', genomes, 'm.self_bsn.npy', 'm.old_prediction.npz', c0, c1, c2, c3, True, params, genomes_per_round=n_genomes, timing=timing)
        with contextlib.redirect_stderr(io.StringIO()):
            for _ in range(warmup):
                step()
            mark0 = len(search_log) if search_log is not None else 0
            t0 = time.perf_counter()
            for _ in range(steps):
                groups, rows = step()
            dt = time.perf_counter() - t0
            mark1 = len(search_log) if search_log is not None else 0
            dt_stores = None
            if with_stores:
                step_with_stores()
                t0 = time.perf_counter()
                for _ in range(steps):
                    step_with_stores()
                dt_stores = time.perf_counter() - t0
    finally:
        os.chdir(cwd)
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    return dict(seconds=dt, seconds_with_stores=dt_stores, genomes=n_genomes * steps, genome_nt=nt, groups_per_step=groups, hit_rows_per_step=rows,
                search_log_timed=(mark0, mark1))          # the slice of search_log the timed plain steps wrote


def _container_cpu():
    """(CPU seconds used by this container so far, periods in which it was throttled) from the control group, or None"""
    try:
        d = dict(l.split() for l in open('/sys/fs/cgroup/cpu.stat'))
        return int(d['usage_usec']) / 1e6, int(d.get('nr_throttled', 0))
    except (OSError, KeyError, ValueError):
        return None


_BUSY_SAMPLER = r"""
import glob, sys, time
import os
mine = [os.path.basename(r) for r in sorted(glob.glob('/dev/dri/renderD*')) if os.access(r, os.R_OK | os.W_OK)]        # the devices this job may open
paths = []
for r in mine:                                  # (render minor 128 + k belongs to card k; other jobs' cards are visible in sysfs too and are not this job's business)
    for p in ('/sys/class/drm/%s/device/gpu_busy_percent' % r, '/sys/class/drm/card%d/device/gpu_busy_percent' % (int(r[7:]) - 128)):
        if os.path.exists(p):
            paths.append(p)
            break
out = open(sys.argv[1], 'w')
out.write('# ' + ' '.join(paths) + '\n')
while True:
    vals = []
    for p in paths:
        try:
            vals.append(open(p).read().strip())
        except Exception:
            vals.append('-1')
    out.write('%.4f %s\n' % (time.time(), ' '.join(vals)))
    out.flush()
    time.sleep(0.01)
"""


class _GpuBusy(object):
    """how busy the GPU is over an interval: the driver's gpu_busy_percent (sysfs) sampled every 10 ms by a process of its own (this process's threads
    are what is being measured).  Only the render nodes this job may open are sampled; of several, the one that was busiest over the interval counts."""

    def __enter__(self):
        import subprocess, tempfile
        self.path = tempfile.mktemp(prefix='pep_busy_')
        try:
            self.proc = subprocess.Popen([sys.executable, '-c', _BUSY_SAMPLER, self.path], stdin=subprocess.DEVNULL, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        except Exception:
            self.proc = None
        self.frac = self.samples = None
        return self

    def mark(self):
        self.t0 = time.time()

    def __exit__(self, *exc):
        t1 = time.time()
        if self.proc is None:
            return False
        self.proc.terminate()
        try:
            self.proc.wait(5)
        except Exception:
            self.proc.kill()
        try:
            rows = [l.split() for l in open(self.path) if not l.startswith('#')]
            rows = [r for r in rows if len(r) > 1 and self.t0 <= float(r[0]) <= t1]
            if rows:
                cols = list(zip(*[[max(0, int(x)) for x in r[1:]] for r in rows]))
                best = max(cols, key=sum)
                self.frac, self.samples = sum(best) / (100.0 * len(best)), len(best)
        except Exception:
            pass
        finally:
            try:
                os.remove(self.path)
            except OSError:
                pass
        return False


def map_strong(args, rank, world, local_rank, n_total, workers=0, warm=False):
    """BASELINE configs[3] / [4], mapping stage, as STRONG scaling: ONE fixed set of n_total synthetic genomes against the exemplar genes through
    get_map_bsn (PEPPAN.py:907-989) - the genomes dealt to the ranks in blocks of 32 (each rank searches its blocks on its own GPU: batched search of
    both tools, -f / -m / -O, K7, K12, build_groups), rank 0 gathers the per-genome columns and writes the four stores.  No data-path collective.
    `workers` > 1: every rank deals its genomes to that many worker processes on its GPU (peppan_amd.mapworkers, the reference's pool of forked
    workers, PEPPAN.py:922) and only keeps the stores; the pool's start-up is inside the measured time unless `warm` (then the set is mapped twice
    by the same pool and the second pass is the one reported, with the first as `first_pass_s`)."""
    import contextlib
    import io
    import tempfile
    from peppan_amd import mapbsn, synth, uberBlast as UB
    names, seqs = synth.make_genes(args.genes, 0, seed=355)
    params = dict(noDiamond=False, match_identity=0.65, match_frag_len=50, match_frag_prop=0.25, link_gap=600, link_diff=1.5, gtable=11,
                  match_len=250., match_len1=100., match_len2=400., match_prop=0.5, match_prop1=0.8, match_prop2=0.4)
    tmp = tempfile.mkdtemp(prefix='pep_maps_%d_' % rank)
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        with open('m.clust.exemplar', 'w') as f:
            for i, s in enumerate(seqs):
                f.write('>%d\n%s\n' % (i, s.decode()))
        worlds = synth.make_genomes(seqs, n_total, seed=355, presence=getattr(args, 'presence', None))                # the same set on every rank
        genomes, nt = {}, 0
        with mapbsn.MapBsn('m.old_prediction.npz', 'w') as op:
            for g, (gname, contig, ann) in enumerate(worlds):
                genomes[100000 + g] = [900000 + g, contig.decode()]
                nt += len(contig)
                op.save(100000 + g, np.array([[k, a, b, st, 1] for k, a, b, st in ann[::2]], dtype=object))
        np.save('m.self_bsn.npy', np.array([[0, 1, 9000], [4, 5, -2]], dtype=int))
        UB._CTX.clear()
        os.environ['PEPPAN_HIP_DEVICE'] = str(local_rank)
        tm, extra = {}, {}

        def once(pool):
            import torch.distributed as dist
            if rank == 0:
                for k in range(4):                 # (a pass before this one left its stores behind: truncating a gigabyte on open is not part of a mapping)
                    if os.path.exists('s%d.npz' % k):
                        os.remove('s%d.npz' % k)
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
            with contextlib.ExitStack() as es:
                stores = [es.enter_context(mapbsn.MapBsn('s%d.npz' % k, 'w')) for k in range(4)] if rank == 0 else [None] * 4
                if workers > 1 and pool is None:
                    from peppan_amd.mapworkers import MapWorkers
                    pool = es.enter_context(MapWorkers(workers, device=local_rank))
                    extra['workers_startup_s'] = time.perf_counter() - t0
                t_call = time.perf_counter()
                mapbsn.get_map_bsn('m', 'm.clust.exemplar', genomes, 'm.self_bsn.npy', 'm.old_prediction.npz', stores[0], stores[1], stores[2], stores[3], True, params,
                                   genomes_per_round=int(os.environ.get('PEPPAN_BENCH_MAP_ROUND', 32)), timing=tm, workers=pool if workers > 1 else 0)
                t_archives = time.perf_counter()
                tm['call'] = t_archives - t_call
            tm['archives_closed'] = time.perf_counter() - t_archives          # (the four stores' central directories written)
            if world > 1:
                dist.barrier()
            return time.perf_counter() - t0
        with contextlib.redirect_stderr(io.StringIO()):
            if workers > 1 and warm:
                from peppan_amd.mapworkers import MapWorkers
                t0 = time.perf_counter()
                with MapWorkers(workers, device=local_rank) as pool:
                    extra['workers_startup_s'] = time.perf_counter() - t0
                    extra['first_pass_s'] = once(pool)
                    c0, p0 = _container_cpu(), time.process_time()
                    with _GpuBusy() as busy:
                        time.sleep(0.05)                   # (the sampler's start-up)
                        busy.mark()
                        dt = once(pool)
                    c1 = _container_cpu()
                    extra['gpu_busy_frac'], extra['gpu_busy_samples'] = busy.frac, busy.samples
                    extra['keeper_process_cpu_s'] = time.process_time() - p0         # (this process alone, all its threads)
                    extra['keeper_feeders_s'] = dict(pool.spent)
                    if c0 and c1:                      # CPU seconds the whole container (this process, its workers) used over the timed pass; times it hit its allowance
                        extra['container_cpu_s'], extra['throttled_periods'] = c1[0] - c0[0], c1[1] - c0[1]
            else:
                once(None) if args.warmup and '--warmup' in sys.argv else None
                dt = once(None)
    finally:
        os.chdir(cwd)
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    return dict(seconds=dt, genomes=n_total, genome_nt=nt, phase_s_rank0=tm, workers=workers, **extra)


def configs4_workloads(local_rank, min_id, min_qcov, torch, steps=5):
    """BASELINE configs[4] at the size ONE GPU holds, behind the headline (the `workloads` block of the line):
      search_50k   the 50 000 x 50 000 gene all-vs-all step (K1 .. K10, the headline's step at 25 x the gene pairs): ms per step, phase split,
                   the three largest kernels with their rooflines (traffic from profiles/r04_counters_50k.json)
      map_50k      50 000 exemplar genes mapped onto 8 genomes of a 50 000-gene pan-genome (6 500 genes, 7 Mb each: synth.PAN_GENOME_PRESENCE)
                   through the batched mapping path - search of both tools, -f / -m / -O, K7, K12, build_groups - and through get_map_bsn with
                   the four stores written; phase split, the time the GPU spends inside the searches, the largest search kernel's roofline"""
    from peppan_amd import _native as N, dist as pdist, synth, uberBlast as UB
    out = {}
    names, seqs = synth.make_genes(50000, 1002, seed=355)
    order = sorted(range(len(names)), key=lambda i: names[i])
    nts = [seqs[i] for i in order]
    params = N.default_params(min_id, min_qcov, 10, 5)
    keys = ('candidates', 'candidates_settled', 'cells', 'cells_settled', 'hits', 'tracebacks', 'tracebacks_gapless', 'seed_hits', 'target_residues', 'query_residues',
            'ms_sw_trace', 'ms_sw', 'ms_seed', 'ms_seed_match', 'ms_trace', 'ms_k1', 'ms_total')
    cyc4 = _profile_tables()[1]
    with N.Context(local_rank) as ctx:
        shard = pdist.ShardedSearch(ctx, nts, nts, params, 0, 1)
        ctx.set_grouping(len(nts), shard.gene_of_target)
        for _ in range(2):
            shard.search(retranslate=True, copy=False)
        torch.cuda.synchronize()
        ctx.set_timing(0)
        t0 = time.perf_counter()
        for _ in range(steps):
            allh, allc, st = shard.search(retranslate=True, copy=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        n_hits, n_clusters = int(len(allh)), int(len(np.unique(ctx.labels)))
        ctx.set_timing(2)
        shard.search(retranslate=True, copy=False)
        acc = dict.fromkeys(keys, 0.0)
        for _ in range(steps):
            st = shard.search(retranslate=True, copy=False)[2]
            for k in keys:
                acc[k] += st[k]
        ctx.set_timing(0)
        per = {k: acc[k] / steps for k in keys}
        rl = roofline_kernels(_profile_tables(PROFILE_COUNTERS_50K)[0], cyc4, os.path.basename(PROFILE_COUNTERS_50K), per, 334, params.n_shapes)
        out['search_50k'] = {'workload': 'synthgenes-v1 seed 355: 50000 genes x 1002 nt, all-vs-all on one GPU (BASELINE configs[4] search stage)', 'steps': steps,
                             'ms_per_step': dt / steps * 1e3, 'gene_pairs_all_vs_all_per_s': 2.5e9 * steps / dt, 'candidates_per_step': per['candidates'], 'hits_per_step': n_hits,
                             'clusters': n_clusters, 'value_gene_pairs_aligned_per_s': per['candidates'] * steps / dt,
                             'sw_cell_updates_per_s_per_gpu': (per['cells'] - per['cells_settled']) / (per['ms_sw'] * 1e-3) if per['ms_sw'] else None,
                             'phase_ms': {k: per[k] for k in ('ms_k1', 'ms_seed', 'ms_seed_match', 'ms_sw', 'ms_sw_trace', 'ms_trace', 'ms_total')},
                             'roofline': rl[0], 'roofline_kernels': rl}
    # ---- the mapping leg: the search context of the drop-in (uberBlast.get_context), every search's statistics recorded
    class A(object):
        genes = 50000
    log, tm = [], {}
    n_genomes = 8
    mr = map_workload(A, 0, 1, local_rank, n_genomes, 1, 1, with_stores=True, gene_set=(names, seqs), presence=synth.PAN_GENOME_PRESENCE, timing=tm, search_log=log)
    # the log holds the searches of: one warm-up step, the timed step, then the steps with stores (whose searches are batched differently: get_map_bsn's own rounds)
    i0, i1 = mr['search_log_timed']
    step_log = log[i0:i1]
    n_per = len(step_log)
    per = {k: float(sum(st[k] for st in step_log)) for k in keys}
    rl = map_step_rooflines(per, step_log, max(1., per['query_residues'] / max(1, n_per) / 50000.), cyc4)
    wall = mr['seconds']
    out['map_50k'] = {'workload': 'synthgenes-v1: 50000 exemplar genes x 1002 nt mapped onto %d genomes of a 50000-gene pan-genome (%.1f Mnt per genome): --blastn --diamond -f -m -O -s 1, K7, K12, '
                                  'build_groups; then the same through get_map_bsn with the four stores (BASELINE configs[4] mapping stage on one GPU)' % (n_genomes, mr['genome_nt'] / n_genomes / 1e6),
                      'genomes_per_s': mr['genomes'] / wall, 'genomes_per_s_with_stores': mr['genomes'] / mr['seconds_with_stores'], 'ms_per_genome': wall / mr['genomes'] * 1e3,
                      'groups_per_genome': mr['groups_per_step'] / n_genomes, 'hit_rows_per_genome': mr['hit_rows_per_step'] / n_genomes,
                      'searches_per_step': n_per, 'gpu_ms_in_searches_per_step': per['ms_total'], 'gpu_busy_frac': per['ms_total'] * 1e-3 / wall,
                      'gpu_busy_note': 'HIP-event time of the searches (K1 .. K8) over the step wall clock; K7 / K11 / K12 run besides (a few per cent more: profiles/r06_map_kernel_stats.txt)',
                      'phase_ms_searches': {k: per[k] for k in ('ms_k1', 'ms_seed', 'ms_seed_match', 'ms_sw', 'ms_sw_trace', 'ms_trace', 'ms_total')},
                      'phase_s_with_stores': tm, 'roofline': rl[0], 'roofline_kernels': rl}
    return out


def north_star_workloads(local_rank, names, order, seqs, nts, min_id, min_qcov, torch, steps=50):
    """The call the reference makes (PEPPAN.py:229-230: uberBlast -r CL -q CL --blastn --diamond -s 1 -e 3,3 ...; uberBlast.py:597-599 runs both
    tools) on the headline's genes, behind the headline:
      search_10k_blastn   the nucleotide tool alone (uberBlast.py:294, 482-509 replaced): the resident nucleotide sets packed as base-code residue sets
                          (both strands of the reference) + the search with exact 17-mers, +2 / -3, gap 6 + 2k, e-value 1e-2 - ms per step, phase split,
                          its three largest kernels with rooflines (counters: profiles/r05_counters_blastn.json)
      north_star_call     both tools back to back through the drop-in's own methods on the NUMERIC table - nucleotide search, translated search,
                          the two tables joined, K7 rescoring (-s 1), fixEnd, the final order - without the object rows uberBlast() builds for its
                          caller (uberblast_e2e_ms has those); what get_similar_pairs consumes"""
    import contextlib
    import io
    import tempfile
    from peppan_amd import _native as N, uberBlast as UB
    out = {}
    keys = ('candidates', 'candidates_settled', 'cells', 'cells_settled', 'hits', 'tracebacks', 'tracebacks_gapless', 'seed_hits', 'target_residues', 'query_residues',
            'ms_sw_trace', 'ms_sw', 'ms_seed', 'ms_seed_match', 'ms_trace', 'ms_k1', 'ms_total')
    cyc4 = _profile_tables()[1]
    pn = N.nucleotide_params(min_id, min_qcov)
    with N.Context(local_rank) as ctx:
        ctx.set_query_nt(nts, 11)
        ctx.set_ref_nt(nts, 6, 11)

        def step():
            ctx.use_nt_as_residues(2)                   # (the nucleotide tool's K1: both sets packed on the device, forward strands + reverse complements)
            return ctx.search(pn, copy=False)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            h, c, st = step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        n_hits = int(len(h))
        ctx.set_timing(2)
        step()
        acc = dict.fromkeys(keys, 0.0)
        for _ in range(steps):
            st = step()[2]
            for k in keys:
                acc[k] += st[k]
        ctx.set_timing(0)
        per = {k: acc[k] / steps for k in keys}
        Lq = per['query_residues'] / max(1, len(nts))
        # (the nucleotide tool's pairs are long: its passes are the <true> launches + the retry launch - kernel families, as for a mapping step)
        rl = family_rooflines(PROFILE_COUNTERS_BLASTN, per, Lq, cyc4, 'one search of the nucleotide tool: both launches of a pass (pairs that fit the staging area / long pairs) summed',
                              'K4a: both strands of the reference streamed through the query index (one shape of weight 17)')
        out['search_10k_blastn'] = {'workload': 'synthgenes-v1 seed 355: %d genes x 1002 nt against themselves, nucleotide tool (the blastn call of uberBlast.py:294), both strands of the reference' % len(nts),
                                    'steps': steps, 'ms_per_step': dt / steps * 1e3, 'candidates_per_step': per['candidates'], 'hits_per_step': n_hits,
                                    'value_gene_pairs_aligned_per_s': per['candidates'] * steps / dt,
                                    'sw_cell_updates_per_s_per_gpu': (per['cells'] - per['cells_settled']) / (per['ms_sw'] * 1e-3) if per['ms_sw'] else None,
                                    'phase_ms': {k: per[k] for k in ('ms_k1', 'ms_seed', 'ms_seed_match', 'ms_sw', 'ms_sw_trace', 'ms_trace', 'ms_total')},
                                    'phase_note': 'ms_k1 is 0 here: the nucleotide sets are packed by nucl_pack inside pep_use_nt_as_residues, in front of the search (inside ms_per_step, outside ms_total)',
                                    'roofline': rl[0], 'roofline_kernels': rl}
    with tempfile.TemporaryDirectory() as tmp:
        fa = os.path.join(tmp, 'exemplar.fa')
        with open(fa, 'w') as f:
            for i in order:
                f.write('>%s\n%s\n' % (names[i], seqs[i].decode()))
        rb = UB.RunBlast()
        rb._as_tables = True
        clock = time.perf_counter

        def call(parts=None, one_by_one=False):
            rb.min_id, rb.min_cov, rb.min_ratio, rb.table_id, rb.n_thread = min_id / 100., 50., min_qcov / 100., 11, 1
            t0 = clock()
            if one_by_one:
                tn = rb._runBlast_table(fa, fa)
                t1 = clock()
                tables = [tn, rb._runDiamond_table(fa, fa)]
            else:
                tables = rb._run_tools(['blastn', 'diamond'], fa, fa, rescore=1)          # (the way run() calls them: side by side, a HIP context each, K7 per tool)
                t1 = t0
            t2 = clock()
            T = rb._post(tables, fa, fa, 1, [False, 0.9, 0.], [False, 300., 1.2], [False, 300, 0.6], [3., 3.], rescored=rb._rescored_by_tools and not one_by_one)
            t3 = clock()
            if parts is not None:
                for k, v in ((('nucleotide_search_ms', t1 - t0), ('translated_search_ms', t2 - t1)) if one_by_one else (('both_tools_side_by_side_with_k7_ms', t2 - t0),)) + (('join_fixend_order_ms' if not one_by_one else 'join_k7_fixend_order_ms', t3 - t2),):
                    parts[k] = parts.get(k, 0.) + v * 1e3
            return T
        with contextlib.redirect_stderr(io.StringIO()):
            call(); call()
            reps, parts, seq_parts, each = 10, {}, {}, []
            t0 = clock()
            for _ in range(reps):
                t_call = clock()
                T = call(parts)
                each.append((clock() - t_call) * 1e3)
            dt = clock() - t0
            call(one_by_one=True)
            t0 = clock()
            for _ in range(reps):
                call(seq_parts, one_by_one=True)
            dt_seq = clock() - t0
        out['north_star_call'] = {'north_star_call_ms': sorted(each)[len(each) // 2], 'mean_ms': dt / reps * 1e3, 'fastest_ms': min(each), 'rows': int(len(T)), 'parts_ms': {k: v / reps for k, v in parts.items()}, 'reps': reps,
                                  'tools_one_after_the_other': {'ms': dt_seq / reps * 1e3, 'parts_ms': {k: v / reps for k, v in seq_parts.items()}},
                                  'what': 'RunBlast: nucleotide tool + translated tool on the %d genes against themselves (FASTA read once and cached; sets resident on the GPU), tables joined, '
                                          'K7 rescoring (-s 1), fixEnd 3,3, final order - the numeric HitTable, no object rows; the median of the calls (a host-bound call on a shared box: one of ten is up to twice the others), mean and fastest beside it' % len(nts)}
    return out


def _ranks_seen(dist, torch, world, share, local_rank):
    """how many ranks the collective library really connects: a sum of ones over the process group - on the GPU through RCCL (backend nccl),
    which is what `rccl_ranks_seen` in the line says; None in a one-rank run without a group"""
    if world == 1:
        return None
    t = torch.ones(1, dtype=torch.int32, device=torch.device('cpu') if share else torch.device('cuda', local_rank))
    dist.all_reduce(t)
    return int(t.item())


VALU_PEAK_CLOCK = 2.4e9          # MI355X peak engine clock; 1024 SIMDs (256 CUs x 4)


class ValuCycles(float):
    """cycles per wave64 VALU instruction and SIMD.  As a number: the packed-16 / DPP / three-operand class (4.16, tools/micro/valu_rate).  of(kernel): weighted by
    the kernel's sweep-loop mix (tools/valu_mix.py) - plain 32-bit VOP1 / VOP2 instructions issue at 2.27, and pricing them at 4.16 put round 5's score pass at
    1.0026 of a 'ceiling'"""
    two = None
    mix = {}

    def of(self, kernel):
        k = self.mix.get(kernel.split('<')[0])
        if not k or not self.two:
            return float(self)
        f2 = float(k.get('two_cycle_frac') or 0.)
        return float(self) * (1. - f2) + float(self.two) * f2


def _kernel_counters(counters, kernel):
    """the tracked PMC record of a kernel; the Smith-Waterman kernels are templates (<false> = the launch of the pairs that fit the LDS staging area,
    <true> = the long pairs): a name without its argument takes whichever instance the profile holds"""
    for name in (kernel, kernel + '<false>', kernel + '<true>'):
        if name in counters:
            return counters[name]
    return {}


def roofline_entry(counters, cyc4, source, kernel, what, ms, alg_bytes, valu=False, scattered=False, launches_per_step=1):
    """One roofline record.  HBM side: algorithmic bytes over the kernel's LIVE duration against the 8 TB/s roof, plus the tracked PMC
    traffic of the same kernel on the same workload (`counters`: per-kernel figures of a rocprofv3 --pmc run kept under profiles/).
    `scattered`: the kernel's reads are 8-byte requests to random lines - its traffic is TCC_EA0_RDREQ x 64 B (the lines that came over the fabric)
    + WRITE_SIZE; the x 2 that MI355X_MICROARCH.md prescribes for FETCH_SIZE is calibrated for wide coalesced streams only and is applied to those.
    `valu`: an integer-VALU bound kernel (the Smith-Waterman passes): the roof that binds is the issue rate of the 4-cycle instruction
    class - `bound` says "valu", `frac` is the issue fraction, the HBM figures move to `hbm`."""
    achieved = alg_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    prof = _kernel_counters(counters, kernel)
    # the tracked counters describe THIS kernel only while it still takes what it took when they were collected: its duration in the
    # PMC passes must agree with the live HIP-event time within 10 %, else the figures are withheld (stale: regenerate with tools/profile_round.sh)
    prof_us = prof.get('avg_us_in_pmc_passes') or []
    stale = bool(prof_us) and ms > 0 and abs(min(prof_us) / 1e3 - ms) > 0.10 * ms and abs(sum(prof_us) / len(prof_us) / 1e3 - ms) > 0.10 * ms
    if stale:
        prof = {}
    traffic = src = None
    if scattered and 'TCC_EA0_RDREQ_sum' in prof and 'WRITE_SIZE' in prof:
        traffic = prof['TCC_EA0_RDREQ_sum'] * 64.0 + prof['WRITE_SIZE'] * 1024.0
        src = ('profiles/%s: TCC_EA0_RDREQ x 64 B + WRITE_SIZE of the same kernel on this workload (separate rocprofv3 --pmc passes); scattered 8-byte '
               'requests, one 64-byte line each - no x 2, which MI355X_MICROARCH.md calibrates for wide coalesced streams' % source)
    elif 'FETCH_SIZE' in prof and 'WRITE_SIZE' in prof:
        traffic = (2 * prof['FETCH_SIZE'] + prof['WRITE_SIZE']) * 1024.0
        src = ('profiles/%s: 2 x FETCH_SIZE + WRITE_SIZE of the same kernel on this workload (separate rocprofv3 --pmc passes; '
               'x2 = the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md for wide coalesced reads)' % source)
    hbm = {'achieved': achieved, 'peak': 8000.0, 'unit': 'GB/s', 'frac': achieved / 8000.0}
    e = {'kernel': kernel, 'what': what, 'bound': 'hbm', 'ms_per_launch': ms, 'launches_per_step': launches_per_step, 'ms_per_step': ms * launches_per_step,
         'algorithmic_bytes': alg_bytes, 'traffic': traffic, 'counters_stale': stale,
         'ms_per_launch_in_pmc_passes': (sum(prof_us) / len(prof_us) / 1e3) if prof_us else None, 'traffic_source': src}
    e.update(hbm)
    if traffic:
        e['traffic_over_algorithmic'] = traffic / alg_bytes if alg_bytes else None
    if valu:
        e['bound'] = 'valu'
        e['hbm'] = hbm
        cyc = cyc4.of(kernel) if isinstance(cyc4, ValuCycles) else cyc4       # (cycles per instruction of THIS kernel's sweep-loop mix)
        peak = 1024 * VALU_PEAK_CLOCK / cyc / 1e9 if cyc else None           # G wave64-instructions / s the chip issues of this mix
        if cyc4 and 'SQ_INSTS_VALU' in prof and ms > 0:
            # wave-instructions per launch (PMC) x measured cycles per instruction of the packed-16 / DPP / add3 class (tools/micro/valu_rate.hip)
            # against 1024 SIMDs at the 2.4 GHz peak clock for the launch's live duration
            rate = prof['SQ_INSTS_VALU'] / (ms * 1e-3) / 1e9
            e.update(achieved=rate, peak=peak, unit='G wave64 VALU instructions/s', frac=rate / peak, valu_issue_frac=rate / peak,
                     valu_issue_source='SQ_INSTS_VALU from profiles/%s x %.3f cycles/instruction (profiles/%s weighted by the sweep loop\'s mix, profiles/%s)' % (
                         source, cyc, os.path.basename(PROFILE_VALU), os.path.basename(PROFILE_VALU_MIX)))
        else:
            e.update(achieved=None, peak=peak, unit='G wave64 VALU instructions/s', frac=None, valu_issue_frac=None,
                     valu_issue_source='no tracked SQ_INSTS_VALU for this kernel on this workload (or stale): the issue fraction is not stated; hbm holds the HBM side')
    return e


def roofline_kernels(counters, cyc4, source, per_step, Lq, n_shapes, seed_weight=10):
    """the three kernels a search spends most of its time in, largest share of the STEP first.  per_step: the search statistics per step (phase
    times in ms, counts).  Algorithmic bytes per launch, SURVEY.md 8(d): SW = sum over pairs of (Lq + Lr) residue bytes + 64 B per reported hit;
    seed join = 1 B + 8 B index entry per target residue + 8 B per raw seed hit"""
    p = per_step
    rl = [roofline_entry(counters, cyc4, source, 'sw_trace_kernel', 'K5 traceback pass: sub-band SW + 4-bit codes over the selected pairs that are not one ungapped run (rule 5a), four per wavefront',
                         p['ms_sw_trace'], (p['tracebacks'] - p['tracebacks_gapless']) * 2 * Lq + p['hits'] * 64, valu=True),
          roofline_entry(counters, cyc4, source, 'sw_score_kernel', 'K5 score pass: banded SW over the candidate pairs that are not identical sequences (those are settled by comparison: candidates_settled)',
                         p['ms_sw'], (p['candidates'] - p['candidates_settled']) * 2 * Lq + p['hits'] * 64, valu=True),
          roofline_entry(counters, cyc4, source, 'seed_match<%d>' % seed_weight, 'K4a: target seeds streamed through the query index (one launch per seed shape)', p['ms_seed_match'] / max(1, n_shapes),
                         9.0 * p['target_residues'] + 8.0 * p['seed_hits'] / max(1, n_shapes), scattered=True, launches_per_step=max(1, n_shapes))]
    rl.sort(key=lambda e: -e['ms_per_step'])
    return rl


FAMILY_MEMBERS = {'sw_trace_kernel': ('sw_trace_kernel', 'sw_trace_retry_kernel'),     # (the pairs that left their sub-band are traced by a launch of their own inside the same phase)
                  'seed_match': ('seed_match', 'seed_match_stride')}                    # (the nucleotide tool's matcher: look-up words at a stride, round 6)


def family_rooflines(path, per, Lq, cyc4, note, seed_what):
    """The kernel FAMILIES a step made of several launches per kernel spends its search time in - all launches of sw_trace_kernel (+ its retry
    launch), of sw_score_kernel, of seed_match whatever their template argument, summed over the step: live HIP-event milliseconds per step from
    the searches' phase timers, algorithmic bytes from their statistics (lengths differ per pair: the mean query length stands in),
    and the PMC traffic of the same launches from the tracked counters at `path` (the same workload under rocprofv3 --pmc: per instance the
    per-dispatch average x its dispatches per step)."""
    prof = json.load(open(path)) if os.path.exists(path) else {}
    kernels, steps = prof.get('kernels', {}), max(1, int(prof.get('steps_per_pass', 1)))
    src = os.path.basename(path)

    def family(prefix, ms_step, alg, what, valu, scattered):
        inst = {k: v for k, v in kernels.items() if k.split('<')[0] in FAMILY_MEMBERS.get(prefix, (prefix,))}
        launches = sum(v.get('dispatches_per_pass', 0) for v in inst.values()) / steps
        pmc_ms = sum(sum(v['avg_us_in_pmc_passes']) / len(v['avg_us_in_pmc_passes']) * v.get('dispatches_per_pass', 0) for v in inst.values() if v.get('avg_us_in_pmc_passes')) / steps / 1e3
        # (the counter passes slow a kernel down by what they collect - the SQ pass most: the fastest pass is the one to hold against the live time, as roofline_entry does)
        pmc_min_ms = sum(min(v['avg_us_in_pmc_passes']) * v.get('dispatches_per_pass', 0) for v in inst.values() if v.get('avg_us_in_pmc_passes')) / steps / 1e3
        # stale = the profile describes other code or another workload: more than 20 % off.  (A mapping step's Smith-Waterman launches run 15 - 18 % longer under the profiler than
        # between the live HIP events - every pass of the set agrees with the others to 2 % -; up to there the counters are used, the issue fraction over the PASS's own duration)
        stale = bool(inst) and ms_step > 0 and abs(pmc_ms - ms_step) > 0.20 * ms_step and abs(pmc_min_ms - ms_step) > 0.20 * ms_step
        traffic = insts = None
        if inst and not stale:
            if scattered and all('TCC_EA0_RDREQ_sum' in v and 'WRITE_SIZE' in v for v in inst.values()):
                traffic = sum((v['TCC_EA0_RDREQ_sum'] * 64.0 + v['WRITE_SIZE'] * 1024.0) * v['dispatches_per_pass'] for v in inst.values()) / steps
            elif all('FETCH_SIZE' in v and 'WRITE_SIZE' in v for v in inst.values()):
                traffic = sum((2 * v['FETCH_SIZE'] + v['WRITE_SIZE']) * 1024.0 * v['dispatches_per_pass'] for v in inst.values()) / steps
            if all('SQ_INSTS_VALU' in v for v in inst.values()):
                insts = sum(v['SQ_INSTS_VALU'] * v['dispatches_per_pass'] for v in inst.values()) / steps
        achieved = alg / (ms_step * 1e-3) / 1e9 if ms_step > 0 else 0.0
        hbm = {'achieved': achieved, 'peak': 8000.0, 'unit': 'GB/s', 'frac': achieved / 8000.0}
        e = {'kernel': prefix, 'instances': sorted(inst) or None, 'what': what, 'bound': 'hbm', 'ms_per_step': ms_step, 'launches_per_step': launches or None,
             'algorithmic_bytes': alg, 'traffic': traffic, 'counters_stale': stale, 'ms_per_step_in_pmc_passes': pmc_ms or None,
             'traffic_source': ('profiles/%s: %s, per instance x dispatches per step' % (src, 'TCC_EA0_RDREQ x 64 B + WRITE_SIZE (scattered requests)' if scattered else '2 x FETCH_SIZE + WRITE_SIZE')) if traffic is not None else None,
             'note': note}
        e.update(hbm)
        if traffic and alg:
            e['traffic_over_algorithmic'] = traffic / alg
        if valu:
            cyc = cyc4.of(prefix) if isinstance(cyc4, ValuCycles) else cyc4
            peak = 1024 * VALU_PEAK_CLOCK / cyc / 1e9 if cyc else None
            basis_ms = pmc_ms if (pmc_ms and abs(pmc_ms - ms_step) > 0.10 * ms_step) else ms_step       # (instructions and duration of the same passes when the live time is another)
            rate = insts / (basis_ms * 1e-3) / 1e9 if insts and basis_ms > 0 else None
            e.update(bound='valu', hbm=hbm, achieved=rate, peak=peak, unit='G wave64 VALU instructions/s', frac=(rate / peak) if rate and peak else None,
                     valu_issue_frac=(rate / peak) if rate and peak else None)
        return e

    rl = [family('sw_trace_kernel', per['ms_sw_trace'], (per['tracebacks'] - per['tracebacks_gapless']) * 2 * Lq + per['hits'] * 64, 'K5 traceback pass (every launch of the step, the retry launch included)', True, False),
          family('sw_score_kernel', per['ms_sw'], (per['candidates'] - per['candidates_settled']) * 2 * Lq + per['hits'] * 64, 'K5 score pass (every launch of the step: pairs that fit the staging area and long pairs)', True, False),
          family('seed_match', per['ms_seed_match'], 9.0 * per['target_residues'] + 8.0 * per['seed_hits'], seed_what, False, True)]
    rl.sort(key=lambda e: -e['ms_per_step'])
    return rl


def map_step_rooflines(per, step_log, Lq, cyc4):
    """family_rooflines over one mapping step (profiles/r05_counters_map50k.json: one mapping step of the same workload under rocprofv3 --pmc)"""
    n_tools = max(1, sum(1 for st in step_log if st['ms_seed_match'] > 0))
    return family_rooflines(PROFILE_COUNTERS_MAP50K, per, Lq, cyc4,
                            'summed over the searches of one mapping step (nucleotide tool + translated tool per sub-batch of genomes); lengths differ per pair: algorithmic bytes use the mean query length',
                            'K4a: the genomes streamed through the exemplar index, both tools (%d searches)' % n_tools)


def _profile_tables(path=None):
    """tracked evidence the line refers to: per-kernel PMC counters of this workload and the measured VALU issue rates"""
    counters, cyc4 = {}, None
    path = path or PROFILE_COUNTERS
    if os.path.exists(path):
        counters = json.load(open(path)).get('kernels', {})
    cyc2 = None
    if os.path.exists(PROFILE_VALU):
        for line in open(PROFILE_VALU):
            f = line.split()
            if len(f) >= 5 and f[0] == 'v_pk_max_i16' and f[1] == '8':
                cyc4 = float(f[2])                     # cycles per wave64 instruction per SIMD at 8 waves/SIMD (packed-16 / DPP / add3 class)
            if len(f) >= 5 and f[0] == 'v_mov_b32' and f[1] == '8':
                cyc2 = float(f[2])                     # ... of plain 32-bit VOP1 / VOP2 instructions
    if cyc4:
        cyc4 = ValuCycles(cyc4)
        cyc4.two = cyc2
        if os.path.exists(PROFILE_VALU_MIX):
            cyc4.mix = json.load(open(PROFILE_VALU_MIX)).get('kernels', {})
    return counters, cyc4


COMPACT_LIMIT = 4096             # bytes of the final stdout line (the driver's parser lost round 5's 25 KB line)


def _short(s, n=118):
    """strings of the compact line stay under the 120 characters the driver keeps of one"""
    if not isinstance(s, str) or len(s) <= n:
        return s
    return s[:n - 3] + '...'


def _num(x, digits=6):
    """a float with `digits` significant digits (None, ints and everything else pass)"""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    if x != x or x in (float('inf'), float('-inf')):
        return None                                  # (strict JSON has no NaN / Infinity)
    return float('%.*g' % (digits, x))


def _get(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def compact_line(detail, detail_file='bench_detail.json'):
    """The ONE line bench.py prints on stdout: the contract's keys, `roofline` of the dominant kernel, `cpu_baseline`, the parity check and one scalar
    per secondary leg - under COMPACT_LIMIT bytes, strict JSON.  Everything else (`workloads`, `roofline_kernels`, notes, sources) is `detail`,
    written to bench_detail.json beside this script."""
    d = detail
    rl, cb = d.get('roofline') or {}, d.get('cpu_baseline') or {}
    line = {k: d.get(k) for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data')}
    cfg = d.get('config') or {}
    line['config'] = {k: _short(v) for k, v in cfg.items()}
    line['sw_cell_updates_per_s_per_gpu'] = d.get('sw_cell_updates_per_s_per_gpu')
    if rl.get('bound') == 'valu':
        # an integer-VALU bound kernel on top (small workloads): the contract's roofline is the HBM one - its HBM side goes into the line, the issue fraction beside it
        rl = dict(rl, bound='hbm', **(rl.get('hbm') or {}))
        line['roofline_valu_issue_frac'] = rl.get('valu_issue_frac')
    line['roofline'] = ({k: _short(rl.get(k), 60) for k in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'algorithmic_bytes', 'traffic_over_algorithmic',
                                                             'ms_per_launch', 'launches_per_step', 'ms_per_step', 'counters_stale')} if rl else None)
    line['cpu_baseline'] = ({k: _short(cb.get(k)) for k in ('value', 'unit', 'cores', 'kind', 'cpu_model', 'sample', 'seconds', 'sw_cells_per_s', 'gpu_hits_identical', 'hits_compared')}
                            if cb else None)
    pc = d.get('parity_check')
    line['parity_check'] = {'gpu_hits_identical': pc.get('gpu_hits_identical'), 'hits_compared': pc.get('hits_compared'), 'against': 'oracle/align_oracle.c'} if pc else None
    # the Smith-Waterman passes beside the dominant kernel: their issue fraction (they are integer-VALU bound)
    for e in d.get('roofline_kernels') or []:
        if e.get('bound') == 'valu':
            line[e['kernel'].replace('_kernel', '') + '_valu_issue_frac'] = e.get('valu_issue_frac')
            line[e['kernel'].replace('_kernel', '') + '_ms'] = e.get('ms_per_step')
    for k in ('hits_per_step', 'candidates_per_step', 'clusters', 'rccl_ranks_seen', 'collective_backend', 'ms_per_step_after_device_sync', 'ms_per_step_incl_h2d',
              'north_star_call_ms', 'uberblast_e2e_ms', 'get_similar_pairs_ms'):
        if d.get(k) is not None:
            line[k] = d[k]
    scal = {
        'get_similar_pairs_search_ms': _get(d, 'get_similar_pairs', 'search_ms'),
        'two_searches_in_flight_ms': _get(d, 'two_searches_in_flight', 'ms_per_step'),
        'search_50k_ms': _get(d, 'workloads', 'search_50k', 'ms_per_step'),
        'search_50k_seed_match_frac': _get(d, 'workloads', 'search_50k', 'roofline', 'frac'),
        'search_10k_blastn_ms': _get(d, 'workloads', 'search_10k_blastn', 'ms_per_step'),
        'map_genomes_per_s': _get(d, 'map_workload', 'genomes_per_s'),
        'pool_genomes_per_s': _get(d, 'map_workload', 'worker_pool', 'genomes_per_s_with_stores'),
        'pool_gpu_busy_frac': _get(d, 'map_workload', 'worker_pool', 'gpu_busy_frac'),
        'pool_cpu_s_per_genome': _get(d, 'map_workload', 'worker_pool', 'cpu_s_per_genome'),
        'map_50k_genomes_per_s': _get(d, 'workloads', 'map_50k', 'genomes_per_s'),
    }
    line.update({k: v for k, v in scal.items() if v is not None})
    err = [k for k in ('workloads', 'map_workload', 'two_searches_in_flight') if _get(d, k, 'error')]
    if err:
        line['leg_errors'] = err
    line['detail'] = detail_file

    def rnd(o):
        if isinstance(o, dict):
            return {k: rnd(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [rnd(v) for v in o]
        return _num(o)
    line = rnd(line)
    text = json.dumps(line, allow_nan=False, separators=(',', ':'))
    # belt and braces: drop the secondary scalars from the end until the line fits
    keep = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline',
            'parity_check', 'sw_cell_updates_per_s_per_gpu')
    while len(text) >= COMPACT_LIMIT:
        extra = [k for k in line if k not in keep]
        if not extra:
            break
        del line[extra[-1]]
        text = json.dumps(line, allow_nan=False, separators=(',', ':'))
    return text


DETAIL_FILE = os.environ.get('PEPPAN_BENCH_DETAIL')          # where the detail record goes instead of beside the script (tests)


def emit(detail):
    """detail -> bench_detail.json (beside the script, and under gpurun_out/ when that exists; or the ONE path PEPPAN_BENCH_DETAIL names); the compact
    line -> stdout, the only line there"""
    def clean(o):
        if isinstance(o, dict):
            return {str(k): clean(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [clean(v) for v in o]
        if isinstance(o, float) and (o != o or o in (float('inf'), float('-inf'))):
            return None
        if isinstance(o, np.generic):
            return clean(o.item())
        return o
    detail = clean(detail)
    for where in ([DETAIL_FILE] if DETAIL_FILE else [os.path.join(ROOT, 'bench_detail.json'), os.path.join(ROOT, 'gpurun_out', 'bench_detail.json')]):
        if os.path.isdir(os.path.dirname(where) or '.'):
            try:
                with open(where, 'w') as f:
                    json.dump(detail, f, indent=1)
                    f.write('\n')
            except OSError as e:
                sys.stderr.write('bench.py: bench_detail.json not written to %s: %r\n' % (where, e))
    sys.stdout.write(compact_line(detail, os.path.basename(DETAIL_FILE) if DETAIL_FILE else 'bench_detail.json') + '\n')
    sys.stdout.flush()


def main_map(args, rank, local_rank, world):
    import torch
    import torch.distributed as dist
    steps = args.steps if '--steps' in sys.argv else 2
    warmup = args.warmup if '--warmup' in sys.argv else 1
    share = os.environ.get('PEPPAN_BENCH_SHARE_GPU') == '1'          # test hook for a one-GPU box (see main): all ranks on device 0, gloo
    if share:
        local_rank = 0
    if world > 1:
        torch.cuda.set_device(local_rank)
        if share:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        dist.barrier()
    torch.cuda.synchronize()
    ranks_seen = _ranks_seen(dist, torch, world, share, local_rank)
    if args.map_scaling == 'strong':
        n_total = args.map_genomes if '--map-genomes' in sys.argv else 500
        r = map_strong(args, rank, world, local_rank, n_total, workers=args.map_workers)
        torch.cuda.synchronize()
        dt = r['seconds']
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=torch.device('cpu') if share else torch.device('cuda', local_rank))
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        if rank == 0:
            print(json.dumps({'metric': 'genomes_mapped_per_s', 'value': r['genomes'] / dt, 'unit': 'genomes/s', 'n_gpus': world, 'steps': 1, 'warmup': 0, 'ms_per_step': dt * 1e3,
                              'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'int16', 'data': 'synthetic', 'rccl_ranks_seen': ranks_seen,
                              'config': {'workload': 'synthgenes-v1: %d exemplar genes (log-normal lengths) x ONE fixed set of %d genomes (%.2f Gnt), get_map_bsn with the four stores written by rank 0 '
                                                     '(BASELINE configs[3]/[4] mapping stage, PEPPAN.py:907-989)' % (args.genes, r['genomes'], r['genome_nt'] / 1e9),
                                         'parallelism': 'genomes dealt to %d rank(s) in blocks of 32, gather_object of the per-genome columns to rank 0, no data-path collective' % world
                                                        + ('; every rank deals its block to %d worker processes on its GPU (start-up inside the time)' % r['workers'] if r['workers'] > 1 else '')},
                              'phase_s_rank0': r['phase_s_rank0'], 'workers_per_rank': r['workers'], 'workers_startup_s': r.get('workers_startup_s'), 'roofline': None, 'cpu_baseline': None}))
            sys.stdout.flush()
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    r = map_workload(args, rank, world, local_rank, args.map_genomes, steps, warmup)
    torch.cuda.synchronize()
    dt = r['seconds']
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=torch.device('cpu') if share else torch.device('cuda', local_rank))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        print(json.dumps({'metric': 'genomes_mapped_per_s', 'value': world * r['genomes'] / dt, 'unit': 'genomes/s', 'n_gpus': world, 'steps': steps, 'warmup': warmup,
                          'ms_per_step': dt / steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'int16', 'data': 'synthetic', 'rccl_ranks_seen': ranks_seen,
                          'config': {'workload': 'synthgenes-v1: %d exemplar genes (log-normal lengths) x %d genomes per rank and step (%.1f Mnt per rank), the genes -> genomes mapping of '
                                                 'BASELINE configs[2..4] (PEPPAN.py:907-989): --blastn --diamond -f -m -O -s 1, build_bsn' % (args.genes, args.map_genomes, r['genome_nt'] / 1e6),
                                     'parallelism': 'genomes sharded over %d rank(s), no data-path collective' % world},
                          'groups_per_step_rank0': r['groups_per_step'], 'hit_rows_per_step_rank0': r['hit_rows_per_step'],
                          'roofline': None, 'cpu_baseline': None,
                          'note': 'secondary workload (the BASELINE metric is --workload search): host-bound - the GPU is busy for a few per cent of a step (DESIGN.md section 4)'}))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=250, help='timed steps (default 250: a timed region of about 0.55 s)')
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--genes', type=int, default=10000)
    ap.add_argument('--gene-len', type=int, default=1002)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-e2e', action='store_true', help='skip the extras after the timed region (round-1 pre-filter comparison loop, H2D-inclusive step, the real uberBlast() call)')
    ap.add_argument('--cpu-sample', type=int, default=10000, help='leading queries of the workload the CPU baseline runs (all host cores)')
    ap.add_argument('--workload', choices=('search', 'map'), default='search', help='search: the all-vs-all step (the BASELINE metric); map: the genes -> genomes '
                    'mapping, genomes sharded over the ranks (weak scaling, no collective) - default 2 timed steps of --map-genomes genomes per rank')
    ap.add_argument('--map-genomes', type=int, default=16, help='genomes per rank and step of --workload map (weak scaling); with --map-scaling strong: the size of the FIXED genome set')
    ap.add_argument('--map-scaling', choices=('weak', 'strong'), default='weak', help='strong: ONE fixed set of --map-genomes genomes (default 500 then) through get_map_bsn, '
                    'sharded over the ranks in blocks, rank 0 writes the four stores (BASELINE configs[3]/[4] mapping stage)')
    ap.add_argument('--map-workers', type=int, default=0, help='--workload map --map-scaling strong: worker processes per rank (peppan_amd.mapworkers; 0 = the rank maps its genomes itself)')
    ap.add_argument('--grid', default=None, help='RxC: query shards x reference shards of the all-vs-all (default: peppan_amd.dist.choose_grid); R*C must equal --gpus')
    ap.add_argument('--no-workloads', action='store_true', help='skip the configs[4] legs behind the headline (50k x 50k search step, 50k-exemplar mapping step)')
    args = ap.parse_args()
    os.environ.setdefault('PEPPAN_LOG', '0')          # the reference-style progress lines of the legs (configure.logger) stay out of the bench's stderr
    if args.gpus > 1 and 'RANK' not in os.environ:
        sys.exit(spawn_ranks(args, sys.argv[1:]))

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.workload == 'map':
        return main_map(args, rank, local_rank, world)
    # the workload, and (rank 0) the CPU oracle run: the cpu_baseline figure at N = 1, the identity check of the gathered table at any N
    # (run first: its OpenMP team is idle while the GPU steps are timed)
    from peppan_amd import synth
    names, seqs = synth.make_genes(args.genes, args.gene_len, seed=355)
    order = sorted(range(len(names)), key=lambda i: names[i])     # FASTA order of the reference: sorted(names) (uberBlast.py:527, 537)
    nts = [seqs[i] for i in order]
    min_id, min_qcov = 45.0, 25.0                                  # PEPPAN.py:229-230 with defaults (match_identity 0.5 - 0.05, match_frag_prop 0.25)
    cpu_line = None
    if rank == 0 and not args.no_cpu_baseline:
        cpu_line = cpu_baseline([names[i] for i in order], nts, min(args.cpu_sample, len(nts)), min_id, min_qcov)

    import torch
    import torch.distributed as dist
    # PEPPAN_BENCH_SHARE_GPU=1 is a test hook for a one-GPU box: all ranks use device 0 and the exchange runs over gloo on
    # host tensors, which exercises every N>1 branch of this file except RCCL itself
    share = os.environ.get('PEPPAN_BENCH_SHARE_GPU') == '1'
    if share:
        local_rank = 0
    if world > 1:
        torch.cuda.set_device(local_rank)
        if share:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    dev = torch.device('cpu') if (share and world > 1) else torch.device('cuda', local_rank)
    ranks_seen = _ranks_seen(dist, torch, world, share, local_rank)

    from peppan_amd import _native as N, dist as pdist
    params = N.default_params(min_id, min_qcov, 10, 5)
    ctx = N.Context(local_rank)
    grid = None
    if args.grid:
        grid = tuple(int(x) for x in args.grid.lower().split('x'))
        if len(grid) != 2 or grid[0] * grid[1] != world:
            sys.exit('bench.py: --grid %s does not describe %d rank(s)' % (args.grid, world))
    shard = pdist.ShardedSearch(ctx, nts, nts, params, rank, world, grid=grid, device=dev if world > 1 else None)

    if world == 1:
        ctx.set_grouping(len(nts), shard.gene_of_target)    # single linkage (K10) as the tail of the search: edges (q, gene of t) straight from the table's device copy

    def step():
        allh, allc, st = shard.search(retranslate=True, copy=False)           # views of the pinned staging area at N = 1: consumed within the step
        if world == 1:
            labels = ctx.labels
        else:
            labels = ctx.components_of_hits(len(nts), allh, shard.gene_of_target)   # the gathered table
        return st, allh, allc, labels

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # host hygiene before anything is timed: with torch imported the interpreter holds a few million objects, and a full collection of
    # the cyclic garbage collector would take milliseconds - more than a step.  Everything allocated so far is moved out of the collector's
    # reach; the steps' own garbage is still collected.
    import gc
    gc.collect()
    gc.freeze()
    for _ in range(args.warmup):
        step()
    sync()
    # phase timers: HIP events on the search's stream cost about 6 us of idle GPU each (profiles/r03_step_timeline.txt), sixteen per search with
    # every phase timed.  The timed region keeps the pair around the dominant kernel (the score pass: the roofline's live duration) and nothing
    # else; the other phases are measured in a loop of their own behind it (phase_timers below)
    ctx.set_timing(1)
    t0 = time.perf_counter()
    keys = ('candidates', 'candidates_settled', 'cells', 'cells_settled', 'cells_swept', 'cells_swept_trace', 'ms_sw_trace', 'ms_sw', 'ms_seed', 'ms_seed_match', 'ms_trace', 'ms_k1', 'ms_total',
            'hits', 'dir_bytes', 'tracebacks', 'tracebacks_gapless', 'seed_hits', 'target_residues', 'query_residues', 'ms_host_translate', 'ms_host_search', 'ms_host_exchange', 'ms_host_merge')
    acc = dict.fromkeys(keys, 0.0)
    t_uf = 0.0
    for _ in range(args.steps):
        t_s = time.perf_counter()
        st, allh, allc, labels = step()
        t_uf += (time.perf_counter() - t_s) * 1e3 - st['ms_host_search'] - st['ms_host_exchange'] - st['ms_host_merge']
        for k in acc:
            acc[k] += st[k]
    sync()
    dt = time.perf_counter() - t0
    # the same K steps once more with every phase timer on (outside the timed region)
    ctx.set_timing(2)
    step()
    sync()
    t_ph = time.perf_counter()
    for _ in range(args.steps):
        st = step()[0]
        for k in ('ms_sw_trace', 'ms_seed', 'ms_seed_match', 'ms_trace', 'ms_k1', 'ms_total'):
            acc[k] += st[k]
    sync()
    ms_step_all_timers = (time.perf_counter() - t_ph) / args.steps * 1e3
    ctx.set_timing(0)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        tot = torch.tensor([acc['candidates'], acc['cells'] - acc['cells_settled']], dtype=torch.float64, device=dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_pairs, total_cells = float(tot[0].item()), float(tot[1].item())
    else:
        total_pairs, total_cells = float(acc['candidates']), float(acc['cells'] - acc['cells_settled'])        # (cells the score pass sweeps)

    parity = None
    if rank == 0 and cpu_line is not None:
        # the oracle's hit table doubles as a check of the GPU's (same queries, same reference): every field, every CIGAR run.  At N > 1 it
        # is the all-gathered, merged table that is checked.
        o_hits, o_cig, n_sample = cpu_line.pop('_hits')
        hits, cig = np.array(allh), np.array(allc)
        g = hits[hits['q'] < n_sample] if n_sample < len(nts) else hits
        same = len(g) == len(o_hits) and all(np.array_equal(g[f], o_hits[f]) for f in ('q', 't', 'q_start', 'q_end', 't_start', 't_end', 'score', 'nm', 'n_ident', 'aln_len', 'cigar_runs'))
        if same and n_sample >= len(nts):
            same = np.array_equal(np.asarray(cig), np.asarray(o_cig))
        parity = dict(gpu_hits_identical=bool(same), hits_compared=int(len(o_hits)), against='oracle C port (align_oracle.c), same queries and reference')
        cpu_line.update(gpu_hits_identical=bool(same), hits_compared=int(len(o_hits)))

    hits_last, clusters_last = float(len(allh)), int(len(np.unique(labels)))      # (allh / labels are views of the context's pinned staging area)
    extras = {}
    if world == 1 and rank == 0:
        # what a caller sees who synchronises the whole device and then searches: the same K steps right behind a torch.cuda.synchronize(),
        # once more (the runtime's one-time 5.6 ms copy-queue stall behind a device-wide synchronisation, DESIGN.md section 6, lands in one of the two loops if it occurs)
        torch.cuda.synchronize()
        t9 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        extras['ms_per_step_after_device_sync'] = (time.perf_counter() - t9) / args.steps * 1e3
    if world == 1 and rank == 0 and not args.no_e2e:
        # Independent searches in flight together (NOT the headline: `value` is one search at a time).  A second context - stream and work space of
        # its own - runs the same step from a second host thread: the seed stage waits for memory (VALU issue 0.2) while the alignment passes wait for
        # nothing but the VALU (0.9), so two searches overlap.  This is what the mapping path's worker processes get out of one GPU.
        try:
            import threading
            ctx2 = N.Context(local_rank)
            shard2 = pdist.ShardedSearch(ctx2, nts, nts, N.default_params(min_id, min_qcov, 10, 5), 0, 1)
            ctx2.set_grouping(len(nts), shard2.gene_of_target)
            for _ in range(3):
                shard2.search(retranslate=True, copy=False)
            n_each = max(1, args.steps // 2)
            go = threading.Barrier(3)

            def runner(sh):
                go.wait()
                for _ in range(n_each):
                    sh.search(retranslate=True, copy=False)
                go.wait()
            th = [threading.Thread(target=runner, args=(sh,)) for sh in (shard, shard2)]
            for t in th:
                t.start()
            torch.cuda.synchronize()
            go.wait()
            t7 = time.perf_counter()
            go.wait()
            torch.cuda.synchronize()
            extras['two_searches_in_flight'] = {'ms_per_step': (time.perf_counter() - t7) / (2 * n_each) * 1e3, 'steps': 2 * n_each, 'contexts': 2,
                                                'note': 'two contexts (streams) on this GPU, one host thread each, the same step; throughput of independent searches, not the headline'}
            for t in th:
                t.join()
            ctx2.close()
        except Exception as e:
            extras['two_searches_in_flight'] = {'error': repr(e)}
        # the SAME unit of work as round 1's line: `value` counts the candidate pairs that enter gapped Smith-Waterman, and since round 2 the
        # ungapped pre-filter in front of it is stricter (threshold 55 instead of 45: 39 % fewer candidates, identical hit table - DESIGN.md
        # section 2).  For comparison across rounds the same timed loop is run once more with the round-1 threshold.
        p45 = N.default_params(min_id, min_qcov, 10, 5, ungapped_min=45)
        p45.stage1_min = 0                                  # (round 1 had no first stage either)
        keep = shard.params
        shard.params = p45
        step(); step()
        torch.cuda.synchronize()
        t5 = time.perf_counter()
        c45 = 0.0
        for _ in range(args.steps):
            c45 += step()[0]['candidates']
        torch.cuda.synchronize()
        d45 = time.perf_counter() - t5
        shard.params = keep
        extras['same_unit_as_round1'] = {'ungapped_min': 45, 'value': c45 / d45, 'unit': 'gene-pairs/s', 'ms_per_step': d45 / args.steps * 1e3,
                                         'candidates_per_step': c45 / args.steps}
    if rank == 0 and world == 1 and not args.no_e2e:
        # (a) the step with the host buffers handed over inside it (H2D of the nucleotides): what the C boundary costs a caller
        reps = max(3, min(args.steps, 10))
        packed = N._pack(nts)                 # what the C boundary takes: one byte buffer + offsets (pep_set_query_nt / pep_set_ref_nt); packing Python strings is not its cost
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(reps):
            ctx.set_query_nt(packed, 11)
            ctx.set_ref_nt(packed, 6, 11)
            step()
        torch.cuda.synchronize()
        extras['ms_per_step_incl_h2d'] = (time.perf_counter() - t1) / reps * 1e3
        t1 = time.perf_counter()
        for _ in range(reps):
            ctx.set_query_nt(nts, 11)
            ctx.set_ref_nt(nts, 6, 11)
            step()
        torch.cuda.synchronize()
        extras['ms_per_step_incl_h2d_from_python_strings'] = (time.perf_counter() - t1) / reps * 1e3
        # (b) the reference's own hot call (PEPPAN.py:229-230) through the drop-in: FASTA in, 16-column object table out -
        # blastn + diamond replacement, -s 1 rescoring (K7), fixEnd, string-keyed sort
        import tempfile
        from peppan_amd import uberBlast as UB
        with tempfile.TemporaryDirectory() as tmp:
            fa = os.path.join(tmp, 'exemplar.fa')
            with open(fa, 'w') as f:
                for i in order:
                    f.write('>%s\n%s\n' % (names[i], seqs[i].decode()))
            argv = ('-r %s -q %s --blastn --diamond -s 1 --min_id 0.45 --min_cov 50 -t 1 --min_ratio 0.25 -e 3,3 -p --gtable 11' % (fa, fa)).split()
            import contextlib
            import io
            with contextlib.redirect_stderr(io.StringIO()):
                UB.uberBlast(argv)                     # warm: FASTA cache, second context, workspaces
                e2e = []
                for _ in range(7):                     # (one call is 20 ms of mostly host work: the median of seven, the fastest beside it)
                    t2 = time.perf_counter()
                    tab = UB.uberBlast(argv)
                    e2e.append((time.perf_counter() - t2) * 1e3)
                extras['uberblast_e2e_ms'], extras['uberblast_e2e_min_ms'] = sorted(e2e)[len(e2e) // 2], min(e2e)
            extras['uberblast_e2e_rows'] = int(tab.shape[0])
            # (c) the consumer of that table, PEPPAN's get_similar_pairs (PEPPAN.py:194-294), through the product's own entry point: numeric table,
            # host scan (C++), get_similar as K14 on the GPU, resolve, exemplar file rewritten - `decide` is everything behind the search
            import shutil
            from peppan_amd import pipeline as PL
            try:
                prio = {int(names[i]): [0, -len(seqs[i]), int(names[i])] for i in order}
            except ValueError:
                prio = None                                     # (gene names that are not integers: PEPPAN encodes them first, PEPPAN.py:1766-1775)
            if prio is not None:
                params_gs = dict(noDiamond=False, match_identity=0.5, match_frag_len=50, n_thread=1, match_frag_prop=0.25, gtable=11, clust_identity=0.9, clust_match_prop=0.8,
                                 incompleteCDS='', match_len=250., match_len1=100., match_len2=400., match_prop=0.5, match_prop1=0.8, match_prop2=0.4)
                ex = os.path.join(tmp, 'p.clust.exemplar')
                np.save(os.path.join(tmp, 'p.clust.npy'), np.zeros((0, 3), dtype=int))
                best = None
                for _ in range(3):
                    shutil.copy(fa, ex)
                    tm = {}
                    with contextlib.redirect_stderr(io.StringIO()):
                        pairs = PL.get_similar_pairs(ex, prio, dict(params_gs, clust=ex), timing=tm)
                    if best is None or tm['decide_ms'] < best['decide_ms']:
                        best = dict(tm, pairs=int(len(pairs)))
                extras['get_similar_pairs_ms'] = best['decide_ms']
                extras['get_similar_pairs'] = {'decide_ms': best['decide_ms'], 'search_ms': best['search_ms'], 'decide_parts_ms': best.get('decide_parts_ms'), 'rows': best['rows'], 'pairs': best['pairs'],
                                               'note': 'decide = classification + host scan + K14 + resolve + exemplar rewrite; search = the uberBlast call in front of it (numeric table, FASTA re-read because the exemplar file was rewritten)'}
        # (d) the genes -> genomes mapping at N = 1 (bench.py --workload map is the same thing per rank): genomes/s on this GPU
        try:
            mr = map_workload(args, 0, 1, local_rank, 16, 1, 1, with_stores=True)
            extras['map_workload'] = {'genomes_per_s': mr['genomes'] / mr['seconds'], 'genomes_per_s_with_stores': mr['genomes'] / mr['seconds_with_stores'],
                                      'genomes': mr['genomes'], 'genome_nt': mr['genome_nt'], 'groups_per_step': mr['groups_per_step'],
                                      'note': 'python bench.py --workload map --gpus N: genomes sharded over the ranks, weak scaling, no collective'}
            # the same path with the reference's pool of workers (PEPPAN.py:922): 8 worker processes share this GPU, this process keeps the stores
            from peppan_amd.configure import effective_cpus
            n_pool, n_set = min(8, max(2, effective_cpus() // 2)), 512          # (BASELINE configs[2]/[3]: 10 000 genes x 500 genomes)
            ps = map_strong(args, 0, 1, local_rank, n_set, workers=n_pool, warm=True)
            extras['map_workload']['worker_pool'] = {'workers': n_pool, 'genomes': n_set, 'genomes_per_s_with_stores': n_set / ps['seconds'], 'seconds': ps['seconds'],
                                                     'first_pass_s': ps['first_pass_s'], 'workers_startup_s': ps['workers_startup_s'], 'phase_s': ps['phase_s_rank0'],
                                                     'container_cpu_s': ps.get('container_cpu_s'), 'cpu_throttled_periods': ps.get('throttled_periods'), 'cpus_granted': effective_cpus(),
                                                     'cpu_s_per_genome': (ps['container_cpu_s'] / n_set) if ps.get('container_cpu_s') else None,
                                                     'gpu_busy_frac': ps.get('gpu_busy_frac'), 'gpu_busy_samples': ps.get('gpu_busy_samples'),
                                                     'gpu_busy_note': "the driver's gpu_busy_percent (sysfs) sampled every 10 ms over the timed pass by a process of its own, setup and archive closes included",
                                                     'rounds_genomes_per_s': n_set / ps['phase_s_rank0']['groups'] if ps['phase_s_rank0'].get('groups') else None,
                                                     'note': 'get_map_bsn(..., workers=8) over ONE set of 512 genomes, four stores written; second pass of a started pool '
                                                             '(first_pass_s: the first one, with every worker\'s first search). python bench.py --workload map --map-scaling strong --map-workers 8'}
        except Exception as e:                                  # never lose the headline over the secondary leg
            extras.setdefault('map_workload', {})['error'] = repr(e)
    if rank == 0 and world == 1 and not args.no_e2e and not args.no_workloads and args.genes == 10000:
        # (e) the call the reference makes - both tools - on the headline's genes; then BASELINE configs[4] at the size one GPU holds: the 50k x 50k
        # search step and a 50k-exemplar mapping step
        try:
            ctx.close()                                         # (the headline's context: its workspaces go back before the large legs allocate theirs)
            ns = north_star_workloads(local_rank, names, order, seqs, nts, min_id, min_qcov, torch)
            extras['north_star_call_ms'] = ns['north_star_call']['north_star_call_ms']
            extras['workloads'] = dict(ns)
            extras['workloads'].update(configs4_workloads(local_rank, min_id, min_qcov, torch))
        except Exception as e:
            import traceback
            extras.setdefault('workloads', {}).update(error=repr(e), traceback=traceback.format_exc()[-1500:])

    if rank == 0:
        K = args.steps
        counters, cyc4 = _profile_tables()
        headline = args.genes == 10000 and args.gene_len == 1002 and world == 1      # the workload the tracked counters were collected on
        Lq = (args.gene_len // 3) if args.gene_len else int(acc['query_residues'] / K / max(1, shard.q1 - shard.q0))
        hits_step = acc['hits'] / K
        n_shapes = params.n_shapes

        rl = roofline_kernels(counters if headline else {}, cyc4, os.path.basename(PROFILE_COUNTERS), {k: acc[k] / K for k in acc}, Lq, n_shapes)
        top = dict(rl[0])
        top['note'] = ('the kernel with the largest share of the STEP (live HIP-event time x launches per step); the Smith-Waterman passes in roofline_kernels are '
                       'integer-VALU bound (SURVEY 8d): their `bound` is "valu", their `frac` the issue fraction, their HBM side sits under `hbm`')
        line = {
            'metric': 'gene_pairs_aligned_per_s', 'value': total_pairs / dt, 'unit': 'gene-pairs/s',
            'n_gpus': world, 'steps': K, 'warmup': args.warmup, 'ms_per_step': dt / K * 1e3,
            'rccl_ranks_seen': ranks_seen, 'collective_backend': None if world == 1 else ('gloo (PEPPAN_BENCH_SHARE_GPU test hook)' if share else 'nccl = RCCL'),
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'int16', 'data': 'synthetic',
            'dtype_note': 'both Smith-Waterman passes run in packed 16-bit integers (v_pk_*_i16) whenever the scores of a pair fit, which is every pair of this workload; 32-bit integer kernels otherwise; seed keys are 64-bit integers',
            'config': {'workload': 'synthgenes-v1 seed 355: %d genes x %d nt, all-vs-all (BASELINE configs[2] search stage), '
                                   'min_id 0.45 min_ratio 0.25 top-k 10 x 5 splits' % (args.genes, args.gene_len),
                       'queries_per_rank': shard.q1 - shard.q0, 'reference_genes_per_rank': shard.g1 - shard.g0,
                       'parallelism': 'grid %d query shards x %d reference shards, one all-gather of the hit tables%s' % (shard.R, shard.C, ' + exact top-k merge' if shard.C > 1 else ' (query shards only: the tables concatenate)')},
            'sw_cell_updates_per_s_per_gpu': (acc['cells'] - acc['cells_settled']) / (acc['ms_sw'] * 1e-3) if acc['ms_sw'] else None,
            'sw_cell_updates_per_s_per_gpu_wall': total_cells / dt / world,
            'hits_per_step': hits_last, 'clusters': clusters_last, 'candidates_per_step': acc['candidates'] / K,
            # units that do not move with internal filters (value counts the candidates the pre-filter lets through):
            'hits_per_s': hits_last * K / dt, 'gene_pairs_all_vs_all_per_s': float(args.genes) * float(args.genes) * K / dt,
            'tracebacks_per_step': acc['tracebacks'] / K, 'tracebacks_gapless_per_step': acc['tracebacks_gapless'] / K,
            'candidates_settled_per_step': acc['candidates_settled'] / K,      # identical pairs: scored by comparison, not swept (their cells are not in the cell rates)
            'value_definition': 'candidate (query, target-frame, band) pairs entering gapped Smith-Waterman per second of step wall time (SURVEY.md 8d-i); '
                                'the pre-filter in front of that stage decides how many there are - see same_unit_as_round1; hits_per_s and '
                                'gene_pairs_all_vs_all_per_s (query genes x reference genes per second) do not depend on it. The timed region starts right behind the barrier + '
                                'torch.cuda.synchronize() the contract asks for (no settle calls in between since round 4); ms_per_step_after_device_sync is a second such loop',
            'phase_ms': {k: acc[k] / K for k in ('ms_k1', 'ms_seed', 'ms_seed_match', 'ms_sw', 'ms_sw_trace', 'ms_trace', 'ms_total')},
            'phase_timers': {'in_timed_region': ['ms_sw'], 'ms_per_step_with_every_phase_timer': ms_step_all_timers,
                             'note': 'pep_set_timing: the timed region records the two HIP events around the score pass (the roofline kernel) only; the other phase_ms '
                                     'entries and the durations of the second and third roofline_kernels come from a loop of the same K steps behind it with all '
                                     'sixteen events per search on - each costs about 6 us of idle GPU between the kernels it separates'},
            'host_phase_ms_rank0': dict({k: acc[k] / K for k in ('ms_host_translate', 'ms_host_search', 'ms_host_exchange', 'ms_host_merge')}, ms_host_union_find=t_uf / K),
            'roofline': top, 'roofline_kernels': rl,
            'cpu_baseline': cpu_line if world == 1 else None,
            'parity_check': parity,
        }
        line.update(extras)
        emit(line)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == '__main__':
    main()
