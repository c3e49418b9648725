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

PROFILE_COUNTERS = os.path.join(ROOT, 'profiles', 'r03_counters.json')     # rocprofv3 PMC passes of this workload (tools/profile_round.sh)
PROFILE_VALU = os.path.join(ROOT, 'profiles', 'r03_valu_rate.txt')         # tools/micro/valu_rate on the same GPU


def cpu_baseline(nts, n_sample, min_id, min_qcov):
    """CPU oracle (C port, OpenMP over targets in the seed phase and over (query, target) pairs in the alignment phase, all host
    cores) timed on a bounded sample of the same workload: the first n_sample queries against the whole reference.
    Reported beside the GPU number, never the target."""
    from oracle import oracle as O
    from peppan_amd.configure import transeq          # numpy translation (pinned to the same golden vectors as the oracle's)
    q_aa = []
    for n, frames in transeq([[i, s.decode()] for i, s in enumerate(nts[:n_sample])], frame='F', transl_table=11):
        q_aa.append(O.aa_codes(min((f[:-1].count('X'), k, f) for k, f in enumerate(frames))[2].replace('-', 'X')))
    t_aa = []
    for n, frames in transeq([[i, s.decode()] for i, s in enumerate(nts)], frame='7', transl_table=11):
        for aa in frames:
            t_aa += [O.aa_codes(c.replace('-', 'X')) for o, c in O.ref_chunks(aa)]
    cores = len(os.sched_getaffinity(0))
    O.lib().oracle_set_threads(cores)
    t0 = time.perf_counter()
    hits, cig, st = O.search(q_aa, t_aa, O.default_params(min_id, min_qcov, 10, 5))
    dt = time.perf_counter() - t0
    return dict(value=st['candidates'] / dt, unit='gene-pairs/s', cores=cores, kind='port',
                sample='first %d of %d queries vs all %d genes x 6 frames; %.1f s on %d threads; %d candidates, %.3g SW cells (%.3g cells/s); '
                       'reference binaries (diamond/blastn/mmseqs) are absent, so this is the oracle C port (scalar code, OpenMP) - its ratio '
                       'to the GPU figure says nothing about DIAMOND'
                       % (n_sample, len(nts), len(nts), dt, cores, st['candidates'], st['cells'], st['cells'] / dt),
                seconds=dt, sw_cells_per_s=st['cells'] / dt, _hits=(hits, cig, n_sample))


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


def map_workload(args, rank, world, local_rank, n_genomes, steps, warmup, with_stores=False):
    """The genes -> genomes mapping (BASELINE configs "x 500 / x 2000 genomes"; PEPPAN.py:907-989) as a bench workload: every rank maps ITS OWN
    n_genomes synthetic genomes (10 000 exemplar genes, 2.2 Mb per genome) - batched GPU search for both tools, -f / -m / -O chain, K7, K12,
    build_bsn - per step.  Genomes are the independent unit of this path: they shard over the ranks with no data-path collective (weak
    scaling), which is what can scale on an 8-GPU node where the 3 ms all-vs-all step cannot."""
    import contextlib
    import io
    import tempfile
    from peppan_amd import mapbsn, synth, uberBlast as UB
    names, seqs = synth.make_genes(args.genes, 0, seed=355)
    params = dict(noDiamond=False, match_identity=0.65, match_frag_len=50, match_frag_prop=0.25, link_gap=600, link_diff=1.5, gtable=11,
                  match_len=250., match_len1=100., match_len2=400., match_prop=0.5, match_prop1=0.8, match_prop2=0.4)
    tmp = tempfile.mkdtemp(prefix='pep_map_%d_' % rank)
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        with open('m.clust.exemplar', 'w') as f:
            for i, s in enumerate(seqs):
                f.write('>%d\n%s\n' % (i, s.decode()))
        worlds = synth.make_genomes(seqs, n_genomes, seed=355 + 1000 * rank)
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
                mapbsn.get_map_bsn('m', 'm.clust.exemplar', genomes, 'm.self_bsn.npy', 'm.old_prediction.npz', c0, c1, c2, c3, True, params, genomes_per_round=n_genomes)
        with contextlib.redirect_stderr(io.StringIO()):
            for _ in range(warmup):
                step()
            t0 = time.perf_counter()
            for _ in range(steps):
                groups, rows = step()
            dt = time.perf_counter() - t0
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
    return dict(seconds=dt, seconds_with_stores=dt_stores, genomes=n_genomes * steps, genome_nt=nt, groups_per_step=groups, hit_rows_per_step=rows)


def _profile_tables():
    """tracked evidence the line refers to: per-kernel PMC counters of this workload and the measured VALU issue rates"""
    counters, cyc4 = {}, None
    if os.path.exists(PROFILE_COUNTERS):
        counters = json.load(open(PROFILE_COUNTERS)).get('kernels', {})
    if os.path.exists(PROFILE_VALU):
        for line in open(PROFILE_VALU):
            f = line.split()
            if len(f) >= 5 and f[0] == 'v_pk_max_i16' and f[1] == '8':
                cyc4 = float(f[2])                     # cycles per wave64 instruction per SIMD at 8 waves/SIMD (packed-16 / DPP / add3 class)
    return counters, cyc4


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
    r = map_workload(args, rank, world, local_rank, args.map_genomes, steps, warmup)
    torch.cuda.synchronize()
    dt = r['seconds']
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=torch.device('cpu') if share else torch.device('cuda', local_rank))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        print(json.dumps({'metric': 'genomes_mapped_per_s', 'value': world * r['genomes'] / dt, 'unit': 'genomes/s', 'n_gpus': world, 'steps': steps, 'warmup': warmup,
                          'ms_per_step': dt / steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'int16', 'data': 'synthetic',
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
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--genes', type=int, default=10000)
    ap.add_argument('--gene-len', type=int, default=1002)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-e2e', action='store_true', help='skip the extras after the timed region (round-1 pre-filter comparison loop, H2D-inclusive step, the real uberBlast() call)')
    ap.add_argument('--cpu-sample', type=int, default=10000, help='leading queries of the workload the CPU baseline runs (all host cores)')
    ap.add_argument('--workload', choices=('search', 'map'), default='search', help='search: the all-vs-all step (the BASELINE metric); map: the genes -> genomes '
                    'mapping, genomes sharded over the ranks (weak scaling, no collective) - default 2 timed steps of --map-genomes genomes per rank')
    ap.add_argument('--map-genomes', type=int, default=16, help='genomes per rank and step of --workload map')
    args = ap.parse_args()
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
        cpu_line = cpu_baseline(nts, min(args.cpu_sample, len(nts)), min_id, min_qcov)

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

    from peppan_amd import _native as N, dist as pdist
    params = N.default_params(min_id, min_qcov, 10, 5)
    ctx = N.Context(local_rank)
    shard = pdist.ShardedSearch(ctx, nts, nts, params, rank, world, device=dev if world > 1 else None)

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
    # After a device-wide synchronisation the HIP runtime re-establishes its SDMA copy queues, once, about a thousand submitted commands
    # later: 5.6 ms in the middle of some step - with twenty timed steps +0.28 ms per step in nine runs out of ten.
    # A burst of tiny SYNCHRONOUS library calls (a union-find over eight nodes each: two uploads, three kernels, one download; 20 ms in all)
    # between the synchronisation and the start of the clock gets the runtime past that point; the device is idle again when the clock
    # starts, and no step is run here.
    settle_a, settle_b = np.array([0, 1, 2, 3], dtype=np.uint32), np.array([1, 2, 3, 4], dtype=np.uint32)
    for _ in range(500):
        ctx.components(8, settle_a, settle_b)
    if world > 1:
        dist.barrier()
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

    extras = {}
    if world == 1 and rank == 0:
        # what a caller sees who synchronises the whole device and then searches: the same K steps right behind a torch.cuda.synchronize(),
        # WITHOUT the settle burst in front - the runtime's one-time 5.6 ms copy-queue stall lands in here if it occurs
        torch.cuda.synchronize()
        t9 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        extras['ms_per_step_after_device_sync'] = (time.perf_counter() - t9) / args.steps * 1e3
    if world == 1 and rank == 0 and not args.no_e2e:
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
                t2 = time.perf_counter()
                tab = UB.uberBlast(argv)
                extras['uberblast_e2e_ms'] = (time.perf_counter() - t2) * 1e3
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
        except Exception as e:                                  # never lose the headline over the secondary leg
            extras['map_workload'] = {'error': repr(e)}

    if rank == 0:
        K = args.steps
        counters, cyc4 = _profile_tables()
        headline = args.genes == 10000 and args.gene_len == 1002 and world == 1      # the workload the tracked counters were collected on
        Lq = (args.gene_len // 3) if args.gene_len else int(acc['query_residues'] / K / max(1, shard.q1 - shard.q0))
        hits_step = acc['hits'] / K
        n_shapes = params.n_shapes

        def entry(kernel, what, ms, alg_bytes, insts_key=None):
            achieved = alg_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            prof = (counters.get(kernel) or counters.get(kernel + '<false>') or {}) if headline else {}      # (the Smith-Waterman kernels are templates: <false> = the launch of the pairs that fit the staging area, all of this workload)
            # the tracked counters describe THIS kernel only while it still takes what it took when they were collected: its duration in the
            # PMC passes must agree with the live HIP-event time within 10 %, else the figures are withheld (stale: regenerate with tools/profile_round.sh)
            prof_us = prof.get('avg_us_in_pmc_passes') or []
            stale = bool(prof_us) and ms > 0 and abs(min(prof_us) / 1e3 - ms) > 0.10 * ms and abs(sum(prof_us) / len(prof_us) / 1e3 - ms) > 0.10 * ms
            if stale:
                prof = {}
            traffic = (2 * prof['FETCH_SIZE'] + prof['WRITE_SIZE']) * 1024.0 if 'FETCH_SIZE' in prof and 'WRITE_SIZE' in prof else None
            e = {'kernel': kernel, 'what': what, 'bound': 'hbm', 'achieved': achieved, 'peak': 8000.0, 'unit': 'GB/s', 'frac': achieved / 8000.0,
                 'ms_per_launch': ms, 'algorithmic_bytes': alg_bytes, 'traffic': traffic, 'counters_stale': stale,
                 'ms_per_launch_in_pmc_passes': (sum(prof_us) / len(prof_us) / 1e3) if prof_us else None,
                 'traffic_source': 'profiles/r03_counters.json: 2 x FETCH_SIZE + WRITE_SIZE of the same kernel on this workload (separate rocprofv3 --pmc passes; '
                                   'x2 = the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md)' if traffic is not None else None}
            if traffic:
                e['traffic_over_algorithmic'] = traffic / alg_bytes if alg_bytes else None
            if insts_key and cyc4 and 'SQ_INSTS_VALU' in prof and ms > 0:
                # integer-VALU bound kernels: wave-instructions per launch (PMC) x measured cycles per instruction of the packed-16 / DPP / add3
                # class (tools/micro/valu_rate.hip) against 1024 SIMDs at the 2.4 GHz peak clock for the launch's live duration
                e['valu_issue_frac'] = prof['SQ_INSTS_VALU'] * cyc4 / (1024 * 2.4e9 * ms * 1e-3)
                e['valu_issue_source'] = 'SQ_INSTS_VALU from profiles/r03_counters.json x %.3f cycles/instruction from profiles/r03_valu_rate.txt' % cyc4
            return e

        ms_sw, ms_tr, ms_match = acc['ms_sw'] / K, acc['ms_sw_trace'] / K, acc['ms_seed_match'] / K / max(1, n_shapes)
        # algorithmic bytes per launch, SURVEY.md 8(d): SW = sum over pairs of (Lq + Lr) residue bytes + 64 B per reported hit;
        # seed join = 1 B + 8 B index entry per target residue + 8 B per raw seed hit
        rl = [entry('sw_trace_kernel', 'K5 traceback pass: sub-band SW + 4-bit codes over the selected pairs that are not one ungapped run (rule 5a), four per wavefront', ms_tr, ((acc['tracebacks'] - acc['tracebacks_gapless']) / K) * 2 * Lq + hits_step * 64, 'v'),
              entry('sw_score_kernel', 'K5 score pass: banded SW over the candidate pairs that are not identical sequences (those are settled by comparison: candidates_settled)', ms_sw, ((acc['candidates'] - acc['candidates_settled']) / K) * 2 * Lq + hits_step * 64, 'v'),
              entry('seed_match<10>', 'K4a: target seeds streamed through the query index (one launch per seed shape)', ms_match,
                    9.0 * acc['target_residues'] / K + 8.0 * acc['seed_hits'] / K / max(1, n_shapes))]
        rl.sort(key=lambda e: -e['ms_per_launch'])
        top = dict(rl[0])
        top['note'] = ('dominant kernel by live HIP-event time; the SW passes are integer-VALU bound by construction (SURVEY 8d): valu_issue_frac is '
                       'the figure that bounds them, the HBM fraction is reported because the contract asks for it; roofline_kernels lists the top three')
        line = {
            'metric': 'gene_pairs_aligned_per_s', 'value': total_pairs / dt, 'unit': 'gene-pairs/s',
            'n_gpus': world, 'steps': K, 'warmup': args.warmup, 'ms_per_step': dt / K * 1e3,
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'int16', 'data': 'synthetic',
            'dtype_note': 'both Smith-Waterman passes run in packed 16-bit integers (v_pk_*_i16) whenever the scores of a pair fit, which is every pair of this workload; 32-bit integer kernels otherwise; seed keys are 64-bit integers',
            'config': {'workload': 'synthgenes-v1 seed 355: %d genes x %d nt, all-vs-all (BASELINE configs[2] search stage), '
                                   'min_id 0.45 min_ratio 0.25 top-k 10 x 5 splits' % (args.genes, args.gene_len),
                       'queries_per_rank': shard.q1 - shard.q0, 'reference_genes_per_rank': shard.g1 - shard.g0,
                       'parallelism': 'grid %d query shards x %d reference shards, one all-gather of the hit tables%s' % (shard.R, shard.C, ' + exact top-k merge' if shard.C > 1 else ' (query shards only: the tables concatenate)')},
            'sw_cell_updates_per_s_per_gpu': (acc['cells'] - acc['cells_settled']) / (acc['ms_sw'] * 1e-3) if acc['ms_sw'] else None,
            'sw_cell_updates_per_s_per_gpu_wall': total_cells / dt / world,
            'hits_per_step': float(len(allh)), 'clusters': int(len(np.unique(labels))), 'candidates_per_step': acc['candidates'] / K,
            # units that do not move with internal filters (value counts the candidates the pre-filter lets through):
            'hits_per_s': float(len(allh)) * K / dt, 'gene_pairs_all_vs_all_per_s': float(args.genes) * float(args.genes) * K / dt,
            'tracebacks_per_step': acc['tracebacks'] / K, 'tracebacks_gapless_per_step': acc['tracebacks_gapless'] / K,
            'candidates_settled_per_step': acc['candidates_settled'] / K,      # identical pairs: scored by comparison, not swept (their cells are not in the cell rates)
            'steady_state': True, 'settle_calls': 500,
            'value_definition': 'candidate (query, target-frame, band) pairs entering gapped Smith-Waterman per second of step wall time (SURVEY.md 8d-i); '
                                'the pre-filter in front of that stage decides how many there are - see same_unit_as_round1; hits_per_s and '
                                'gene_pairs_all_vs_all_per_s (query genes x reference genes per second) do not depend on it. ms_per_step is a steady-state figure: '
                                '500 tiny library calls run between the device-wide synchronisation and the clock (settle_calls); ms_per_step_after_device_sync is the same loop without them',
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
        print(json.dumps(line))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == '__main__':
    main()
