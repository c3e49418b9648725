"""The multi-GPU exchange step with the hit table left on the GPU, on a one-rank RCCL group (all a one-GPU box offers): checks that
Context.search_on_device -> dist.allgather_hits(on_device=...) returns the table the host-staged exchange returns, and times both.
Run by tests/test_gpu_parity.py::test_allgather_from_device_memory_over_rccl.   python3 tools/exchange_device_check.py [n_genes]"""
import os
import sys
import time

import numpy as np
import torch                          # before the library: the torch wheel carries its own HIP runtime, which must come up first
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
    from peppan_amd import _native as N, synth, dist as pdist
    n_genes = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29541')
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    ctx = N.Context(0)
    try:
        names, seqs = synth.make_genes(n_genes, 0 if n_genes < 5000 else 1002, seed=8 if n_genes < 5000 else 355)
        ctx.set_query_nt(seqs, 11)
        ctx.set_ref_nt(seqs, 6, 11)
        p = N.default_params(45., 25., 10, 5)
        h, c, st = ctx.search(p)
        want = h.copy()
        want['q'] += 5
        want['t'] += 7
        hint, hint2 = {}, {}
        t_dev = t_host = 0.
        for rep in range(6):                                  # first call: two collectives (sizes, payload); then one (size hint)
            t0 = time.perf_counter()
            nh, nc, st2, ptrs = ctx.search_on_device(p)
            t1 = time.perf_counter()
            gh, gc = pdist.allgather_hits(None, None, 5, 7, device=dev, hint=hint, on_device=(nh, nc) + tuple(ptrs), force=True)
            t2 = time.perf_counter()
            assert (nh, nc) == (len(h), len(c)) and ptrs[0] and ptrs[1]
            assert np.array_equal(gh, want) and np.array_equal(gc, c)
            t3 = time.perf_counter()
            h2, c2, st3 = ctx.search(p, copy=False)
            t4 = time.perf_counter()
            hh, hc = pdist.allgather_hits(h2, c2, 5, 7, device=dev, hint=hint2, force=True)                     # the host-staged path
            t5 = time.perf_counter()
            assert np.array_equal(hh, want) and np.array_equal(hc, c)
            if rep >= 2:
                t_dev += (t2 - t0) / 4
                t_host += (t5 - t3) / 4
        assert hint.get('slot', 0) >= 8 + len(h) * 64 + 4 * len(c)
        print('search + exchange of %d hits, table left on the device: %.3f ms; through the host: %.3f ms' % (len(h), t_dev * 1e3, t_host * 1e3))
        # the table is still on the device: K10 reads it there, and the host copy can be had after all
        nh, nc, st2, ptrs = ctx.search_on_device(p)
        lab = ctx.components_of_search(len(seqs), ctx.target_meta()['seq'])
        assert np.array_equal(lab, ctx.components_of_hits(len(seqs), h, ctx.target_meta()['seq']))
        fh, fc = ctx.result_to_host()
        assert np.array_equal(fh, h) and np.array_equal(fc, c)
        # an empty table travels too
        ctx.set_query_nt([b'ACGT' * 30], 11)
        nh, nc, st3, ptrs = ctx.search_on_device(p)
        gh, gc = pdist.allgather_hits(None, None, 0, 0, device=dev, hint={}, on_device=(nh, nc) + tuple(ptrs), force=True)
        assert nh == 0 and len(gh) == 0 and len(gc) == 0
        print('exchange from device memory: ok')
    finally:
        ctx.close()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
