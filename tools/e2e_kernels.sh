#!/bin/bash
# kernel table of the reference's hot call through the drop-in (tools/e2e_profile.py: uberBlast --blastn --diamond -s 1 on 10 k genes, four calls)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats -d gpurun_out/e2e -o e2e -- python3 tools/e2e_profile.py > gpurun_out/e2e.log 2>&1
python3 tools/rocpd_summary.py gpurun_out/e2e/e2e_results.db > gpurun_out/e2e_stats.txt
head -30 gpurun_out/e2e_stats.txt; grep "uberBlast:" gpurun_out/e2e.log; rm -rf gpurun_out/e2e
