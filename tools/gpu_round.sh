#!/bin/bash
# one GPU round trip while iterating: parity tests, the traced step, the consumer timing, kernel table at 50 k genes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -4
bash tools/trace_step.sh > /dev/null 2>&1
grep -E "seed_runs|seed_extend|seed_match|idx_|step span" gpurun_out/t2_gaps.txt | head -12
timeout 300 python3 tools/similar_timing.py > gpurun_out/similar_timing.txt 2>&1; head -4 gpurun_out/similar_timing.txt
if [ "$1" = "50k" ]; then
  rocprofv3 --kernel-trace --stats -d gpurun_out/t50 -o t50 -- python3 bench.py --genes 50000 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > gpurun_out/t50.log 2>&1
  python3 tools/rocpd_summary.py gpurun_out/t50/t50_results.db > gpurun_out/t50_stats.txt; head -16 gpurun_out/t50_stats.txt; rm -rf gpurun_out/t50
fi
timeout 200 python bench.py --no-cpu-baseline --no-e2e 2>/dev/null | head -c 330
