"""Where the interpreter time of the genes -> genomes mapping goes, thread by thread: python tools/map_hot_spots.py [genomes per step] [steps]
bench.map_workload (one process: search thread, groups on the caller's thread, stores thread) under cProfile, every thread with a profile of its own."""
import argparse, cProfile, io, os, pstats, sys, threading, time
sys.path.insert(0, '.')
os.environ.setdefault('PEPPAN_LOG', '0')
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
profiles = {}
plain_run = threading.Thread.run


def run(self):
    pr = cProfile.Profile()
    profiles[self.name] = pr
    pr.enable()
    try:
        plain_run(self)
    finally:
        pr.disable()


args = argparse.Namespace(genes=10000, warmup=1, presence=None)
bench.map_workload(args, 0, 1, 0, n, 1, 1, with_stores=True)          # warm: contexts, caches
threading.Thread.run = run
main = cProfile.Profile()
t0 = time.perf_counter()
main.enable()
r = bench.map_workload(args, 0, 1, 0, n, steps, 0, with_stores=True)
main.disable()
dt = time.perf_counter() - t0
threading.Thread.run = plain_run
print('%d genomes x %d steps: %.1f genomes/s without stores, %.1f with; whole call %.2f s' % (n, steps, r['genomes'] / r['seconds'], r['genomes'] / r['seconds_with_stores'], dt))
profiles['MAIN'] = main
for name, pr in sorted(profiles.items(), key=lambda kv: -sum(v[2] for v in pstats.Stats(kv[1]).stats.values())):
    st = pstats.Stats(pr)
    tot = sum(v[2] for v in st.stats.values())
    if tot < 0.02:
        continue
    out = io.StringIO()
    st.stream = out
    st.sort_stats('tottime').print_stats(14)
    lines = [l for l in out.getvalue().splitlines() if l.strip()]
    print('---- thread %s: %.2f s of profiled time (%.1f ms per genome)' % (name, tot, tot / (n * steps) * 1e3))
    for l in lines[4:22]:
        print('   ' + l[:170])
