cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for d in 0 1; do
  rocprofv3 --kernel-trace --stats -d gpurun_out/cf$d -o cf$d -- python3 tools/ab/cell_floor.py $d > gpurun_out/cf$d.log 2>&1
  python3 tools/rocpd_summary.py gpurun_out/cf$d/cf${d}_results.db > gpurun_out/cf${d}_stats.txt; echo "debug $d"; head -14 gpurun_out/cf${d}_stats.txt; tail -1 gpurun_out/cf$d.log; rm -rf gpurun_out/cf$d
done
