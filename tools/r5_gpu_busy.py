"""round 5: how busy the GPU is while the worker pool maps (gpu_busy_percent of the driver, sampled every 10 ms by a process of its own)
python tools/r5_gpu_busy.py <seconds> <out file>"""
import glob, sys, time
paths = glob.glob('/sys/class/drm/card*/device/gpu_busy_percent')
secs, out = float(sys.argv[1]), sys.argv[2]
t0, rows = time.time(), []
while time.time() - t0 < secs:
    vals = []
    for p in paths:
        try:
            vals.append(int(open(p).read()))
        except Exception:
            vals.append(-1)
    rows.append((time.time(), vals))
    time.sleep(0.01)
with open(out, 'w') as f:
    f.write('# %s\n' % paths)
    for t, v in rows:
        f.write('%.3f %s\n' % (t, ' '.join(map(str, v))))
