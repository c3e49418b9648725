"""What one exchange step costs on a GPU box besides the wire: the collectives of peppan_amd.dist._all_gather_bytes on a
one-rank RCCL group (staging copies, launches, synchronisation - everything except the xGMI transfer itself) and the host merge
(pep_merge_hits) of a table of the bench workload's size.  python3 tools/exchange_cost.py [n_hits]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
    import torch
    import torch.distributed as dist
    from peppan_amd import _native as N, dist as pdist
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 36000
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    rng = np.random.default_rng(1)
    h = np.zeros(n, dtype=N.HIT_DTYPE)
    h['q'] = np.sort(rng.integers(0, 10000, n))
    h['t'] = rng.integers(0, 60000, n)
    h['score'] = rng.integers(50, 900, n)
    h['cigar_runs'] = 3
    h['cigar_off'] = np.arange(n) * 3
    a, b = h[h['t'] < 30000], h[h['t'] >= 30000]
    a, b = a[np.lexsort((a['t'], a['q']))], b[np.lexsort((b['t'], b['q']))]
    allh, cig = np.concatenate([a, b]), np.arange(3 * n, dtype=np.uint32)
    raw = np.concatenate([allh.view(np.uint8), cig.view(np.uint8)])

    def fill(view):
        view[:] = raw
    hint = {}
    for name, fn in (('all-gather of %.2f MB through the staging buffers (1-rank RCCL group), counts + payload' % (raw.size / 1e6), lambda: pdist._all_gather_bytes(fill, raw.size, None, dev)),
                     ('the same with a size hint: one collective', lambda: pdist._all_gather_bytes(fill, raw.size, None, dev, hint=hint)),
                     ('pep_merge_hits of %d hits from 2 reference shards' % n, lambda: N.merge_hits(allh, cig, 10, 5, out=keep))):
        keep = {}
        for _ in range(3):
            fn()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        print('%-80s %.3f ms' % (name, (time.perf_counter() - t0) / 20 * 1e3))
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
