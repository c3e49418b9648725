#!/bin/bash
# round-4 starting point on one box: kernel tables of the bench step at 10 k and 50 k genes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for g in 10000 50000; do
  rocprofv3 --kernel-trace --stats -d gpurun_out/b$g -o b$g -- python3 bench.py --genes $g --steps 5 --warmup 2 --no-cpu-baseline --no-e2e > gpurun_out/b$g.log 2>&1
  python3 tools/rocpd_summary.py gpurun_out/b$g/b${g}_results.db > gpurun_out/b${g}_stats.txt; rm -rf gpurun_out/b$g
  head -24 gpurun_out/b${g}_stats.txt
done
