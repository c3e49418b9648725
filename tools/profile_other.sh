#!/bin/bash
# Runs ON THE GPU BOX: durations + FETCH_SIZE / WRITE_SIZE of the kernels outside the bench step (K7, K9, K11, K12, K13) on config-size inputs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats -d gpurun_out/oth -o oth -- python3 tools/other_kernels.py > gpurun_out/other_kernels.log 2>&1
python3 tools/rocpd_summary.py gpurun_out/oth/oth_results.db > gpurun_out/oth_stats.txt 2> gpurun_out/oth_stats.err; ls -la gpurun_out/oth >> gpurun_out/oth_stats.err
bash tools/pmc_run.sh oth_f FETCH_SIZE -- tools/other_kernels.py
bash tools/pmc_run.sh oth_w WRITE_SIZE -- tools/other_kernels.py
python3 tools/pmc_to_json.py gpurun_out/other_counters.json 2>> gpurun_out/oth_stats.err gpurun_out/oth_f/oth_f_results.db gpurun_out/oth_w/oth_w_results.db
cat gpurun_out/other_kernels.log | tail -8
