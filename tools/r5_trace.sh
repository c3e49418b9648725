#!/bin/bash
# round 5: kernel table of a few searches at $1 genes (default 10000), self-search shortcut on (default) or off ($2 = 8)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
N=${1:-10000}; OFF=${2:-0}; TAG=${3:-r5t}
rocprofv3 --kernel-trace --stats -d gpurun_out/$TAG -o $TAG -- python3 tools/one_search.py $N $OFF > gpurun_out/$TAG.log 2>&1
python3 tools/rocpd_summary.py gpurun_out/$TAG/${TAG}_results.db > gpurun_out/${TAG}_stats.txt
head -28 gpurun_out/${TAG}_stats.txt | cut -c1-150
