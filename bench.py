#!/usr/bin/env python3
"""bench.py - one "step" = one full pass of the hot path over the 10k-gene all-vs-all workload:
K1 translate/pack -> K2-K4 seeds/candidates -> K5 banded Smith-Waterman -> K6 traceback -> K8 filters/top-k ->
hit table to the host -> (N>1: RCCL all-gather of the shard hit tables) -> K10 union-find.
Inputs (nucleotides) are resident in HBM before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--genes G] [--no-cpu-baseline]

N>1 is launched by torch.distributed.run, one rank per GPU; queries are sharded, the reference replicated
(strong scaling: the total workload is the named 10k x 10k configuration whatever N is)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


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
                       'reference binaries (diamond/blastn/mmseqs) are absent, so this is the oracle C port (scalar code, OpenMP)'
                       % (n_sample, len(nts), len(nts), dt, cores, st['candidates'], st['cells'], st['cells'] / dt),
                seconds=dt, sw_cells_per_s=st['cells'] / dt, _hits=(hits, cig, n_sample))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--genes', type=int, default=10000)
    ap.add_argument('--gene-len', type=int, default=1002)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample', type=int, default=10000, help='leading queries of the workload the CPU baseline runs (all host cores)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    # the workload, and (rank 0, N=1 only) the CPU baseline (run first: its OpenMP team is idle while the GPU steps are timed)
    from peppan_amd import synth
    names, seqs = synth.make_genes(args.genes, args.gene_len, seed=355)
    order = sorted(range(len(names)), key=lambda i: names[i])     # FASTA order of the reference: sorted(names) (uberBlast.py:527, 537)
    nts = [seqs[i] for i in order]
    min_id, min_qcov = 45.0, 25.0                                  # PEPPAN.py:229-230 with defaults (match_identity 0.5 - 0.05, match_frag_prop 0.25)
    cpu_line = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
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
    bounds = pdist.shard_bounds([len(s) for s in nts], world)
    q0, q1 = bounds[rank], bounds[rank + 1]
    params = N.default_params(min_id, min_qcov, 10, 5)

    ctx = N.Context(local_rank)
    ctx.set_query_nt(nts[q0:q1], 11)
    ctx.set_ref_nt(nts, 6, 11)
    gene_of_target = None

    def step():
        nonlocal gene_of_target
        ctx.translate(force=True)
        hits, cig, st = ctx.search(params, copy=False)          # views of the pinned staging area: consumed within the step
        if gene_of_target is None:
            gene_of_target = ctx.target_meta()['seq'].astype(np.uint32)
        allh, allc = pdist.allgather_hits(hits, cig, q0, device=dev if world > 1 else None)
        labels = ctx.components_of_hits(len(nts), allh, gene_of_target)          # edges (q, gene of t) straight from the table
        return hits, st, allh, labels

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    acc = dict(candidates=0, cells=0, cells_swept=0, ms_sw_trace=0.0, ms_sw=0.0, ms_seed=0.0, ms_trace=0.0, ms_k1=0.0, ms_total=0.0, hits=0, dir_bytes=0, tracebacks=0)
    for _ in range(args.steps):
        hits, st, allh, labels = step()
        for k in acc:
            acc[k] += st[k]
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        tot = torch.tensor([acc['candidates'], acc['cells']], dtype=torch.float64, device=dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_pairs, total_cells = float(tot[0].item()), float(tot[1].item())
    else:
        total_pairs, total_cells = float(acc['candidates']), float(acc['cells'])

    if rank == 0 and cpu_line is not None:
        # the oracle's hit table of the baseline run doubles as a check of the GPU's (same queries, same reference): every field, every CIGAR run
        o_hits, o_cig, n_sample = cpu_line.pop('_hits')
        hits, cig, _ = ctx.search(params)           # (one more search, outside the timed region: a private copy of hits + CIGARs)
        g = hits[hits['q'] < n_sample] if n_sample < len(nts) else hits
        same = len(g) == len(o_hits) and all(np.array_equal(g[f], o_hits[f]) for f in ('q', 't', 'q_start', 'q_end', 't_start', 't_end', 'score', 'nm', 'n_ident', 'aln_len', 'cigar_runs'))
        if same and n_sample >= len(nts):
            same = np.array_equal(np.asarray(cig), np.asarray(o_cig))
        cpu_line['gpu_hits_identical'] = bool(same)
        cpu_line['hits_compared'] = int(len(o_hits))
    if rank == 0:
        K = args.steps
        cand = acc['candidates'] / K
        ms_sw = acc['ms_sw'] / K
        # dominant kernel = K5 sw_kernel (one launch per step).  Algorithmic bytes per launch, SURVEY.md 8(d):
        # sum over candidate pairs of (Lq + Lr) residue bytes + 64 B per reported hit.
        Lq = args.gene_len // 3
        alg_bytes = cand * (2 * Lq) + (acc['hits'] / K) * 64
        achieved = alg_bytes / (ms_sw * 1e-3) / 1e9
        traffic, traffic_note = None, ''
        tfile = os.path.join(ROOT, 'profiles', 'r01_traffic.json')
        if os.path.exists(tfile) and args.genes == 10000 and args.gene_len == 1002 and world == 1:
            k = json.load(open(tfile))['kernels'].get('sw_score_kernel')
            if k:
                # rocprofv3 PMC passes of the same workload (profiles/r01_pmc_hbm_traffic.txt): FETCH_SIZE is doubled per the gfx950
                # correction of MI355X_MICROARCH.md (HBM section); WRITE_SIZE matched the known output bytes exactly
                traffic = (2 * k['FETCH_SIZE'] + k['WRITE_SIZE']) * 1024.0
                traffic_note = '; PMC traffic = 2 x FETCH_SIZE + WRITE_SIZE from profiles/r01_pmc_hbm_traffic.txt (below the algorithmic bytes: residues shared by candidates are served from L2)'
        line = {
            'metric': 'gene_pairs_aligned_per_s', 'value': total_pairs / dt, 'unit': 'gene-pairs/s',
            'n_gpus': world, 'steps': K, 'warmup': args.warmup, 'ms_per_step': dt / K * 1e3,
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'int32', 'data': 'synthetic',
            'config': {'workload': 'synthgenes-v1 seed 355: %d genes x %d nt, all-vs-all (BASELINE configs[2] search stage), '
                                   'min_id 0.45 min_ratio 0.25 top-k 10 x 5 splits' % (args.genes, args.gene_len),
                       'queries_per_rank': q1 - q0, 'parallelism': 'query-shard x%d, reference replicated' % world},
            'sw_cell_updates_per_s_per_gpu': acc['cells'] / (acc['ms_sw'] * 1e-3),
            'sw_cell_updates_per_s_per_gpu_wall': total_cells / dt / world,
            'hits_per_step': float(len(allh)), 'clusters': int(len(np.unique(labels))),
            'phase_ms': {k: acc[k] / K for k in ('ms_seed', 'ms_sw', 'ms_sw_trace', 'ms_trace', 'ms_total')},
            'roofline': {'bound': 'hbm', 'kernel': 'sw_score_kernel (K5 banded Smith-Waterman, score pass over all candidate pairs)', 'achieved': achieved, 'peak': 8000.0, 'unit': 'GB/s',
                         'frac': achieved / 8000.0, 'traffic': traffic,
                         # what actually bounds it: 120 VALU instructions per 16 candidate-steps of 64 cells (ISA of sw_score_kernel's
                         # packed 16-bit loop body), each occupying a SIMD for 4 cycles (PMC: SQ_ACTIVE_INST_VALU == SQ_INSTS_VALU
                         # quad-cycles), 1024 SIMDs at 2.4 GHz
                         'valu_issue_frac': (acc['cells_swept'] / K / 64) * (120. / 16.) * 4 / (1024 * 2.4e9 * ms_sw * 1e-3),
                         'note': 'integer-VALU-bound by construction (SURVEY 8d): valu_issue_frac = share of the VALU issue ceiling; the traceback '
                                 'pass sw_kernel<true> (selected pairs only) is a second launch that writes %.3g B of traceback codes in %.2f ms '
                                 '= %.0f GB/s' % (acc['dir_bytes'] / K, acc['ms_sw_trace'] / K, acc['dir_bytes'] / K / (acc['ms_sw_trace'] / K * 1e-3) / 1e9) + traffic_note},
            'cpu_baseline': cpu_line,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()
    ctx.close()


if __name__ == '__main__':
    main()
