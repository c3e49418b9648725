cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats -d gpurun_out/t2 -o t2 -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e > gpurun_out/t2.log 2>&1
python3 tools/rocpd_summary.py gpurun_out/t2/t2_results.db > gpurun_out/t2_stats.txt
python3 tools/rocpd_gaps.py gpurun_out/t2/t2_results.db > gpurun_out/t2_gaps.txt
tail -30 gpurun_out/t2_gaps.txt
