import time, sys, os
sys.path.insert(0, '.')
from peppan_amd.mapworkers import MapWorkers
def t_start(tag):
    t = time.perf_counter()
    with MapWorkers(8) as p:
        print(tag, 'started in %.2f s' % (time.perf_counter() - t), flush=True)
t_start('plain')
big = [bytes(2_200_000) for _ in range(512)]
t_start('1.1 GB of strings')
import torch
torch.cuda.synchronize()
t_start('torch.cuda initialised')
from peppan_amd import _native as N
ctx = N.Context(0)
t_start('own context')
