# A/B of library variants on ONE box: tools/ab/lib_<tag>.so (built by hand with a -D switch) against the tree's library; for each the traced step
# (kernel named in $1) and the host-clock time of 3 x 30 searches of the bench workload, K1 inside
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp peppan_amd/libpeppan_hip.so /tmp/lib_base.so
for v in base "${@:2}" base; do
  if [ $v = base ]; then cp /tmp/lib_base.so peppan_amd/libpeppan_hip.so; else cp tools/ab/lib_$v.so peppan_amd/libpeppan_hip.so; fi
  bash tools/trace_step.sh > /dev/null 2>&1
  echo "== $v: $(grep -E "^$1 " gpurun_out/t2_gaps.txt | head -2 | tr -s ' ' | tr '\n' ';')  $(grep 'step span' gpurun_out/t2_gaps.txt)"
  python3 - <<'PY'
import sys, time
sys.path.insert(0, '.')
from peppan_amd import _native as N, synth
names, seqs = synth.make_genes(10000, 1002, seed=355)
order = sorted(range(len(names)), key=lambda i: names[i])
nts = [seqs[i] for i in order]
ctx = N.Context(0)
p = N.default_params(45., 25., 10, 5)
ctx.set_query_nt(nts, 11); ctx.set_ref_nt(nts, 6, 11)
for _ in range(5):
    ctx.invalidate_translation(); ctx.search(p, copy=False)
ts = []
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(30):
        ctx.invalidate_translation(); h, c, st = ctx.search(p, copy=False)
    ts.append((time.perf_counter() - t0) / 30 * 1e3)
print('   ms per search: ' + ' '.join('%.3f' % t for t in ts), len(h), 'hits')
PY
done
cp /tmp/lib_base.so peppan_amd/libpeppan_hip.so
