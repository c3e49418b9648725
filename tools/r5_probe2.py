import sys, time
sys.path.insert(0, '.')
import numpy as np
from peppan_amd import uberBlast as UB
ctx = UB.get_context()
def probe(tag):
    t = time.perf_counter()
    ctx.overlaps(np.zeros(2, np.int32), np.array([1, 5]), np.array([10, 20]), np.array([0, 1]), 300., 0.6)
    print('   probe %-44s %.2f ms' % (tag, (time.perf_counter() - t) * 1e3))
probe('first'); probe('second')
for mb in (1, 4, 10, 40):
    x = np.empty(mb << 20, np.uint8); x[:] = 1
    probe('after allocating + touching %d MB' % mb); probe('again')
    del x
    probe('after freeing it'); probe('again')
keep = []
for rep in range(3):
    keep.append(np.ones(10 << 20, np.uint8))
    probe('after a 10 MB array that is kept')
y = np.ones(10 << 20, np.uint8)
for rep in range(3):
    y[:] = rep
    probe('after rewriting an old 10 MB array')
