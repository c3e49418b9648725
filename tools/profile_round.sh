#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash tools/profile_round.sh'): collects everything tools/refresh_profiles.sh copies into profiles/.
#   kernel trace + stats of the default bench, the step timeline, FETCH_SIZE / WRITE_SIZE in separate PMC passes, the bench line.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats -d gpurun_out/final -o final -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/final.log 2>&1
python3 tools/rocpd_summary.py gpurun_out/final/final_results.db > gpurun_out/final_stats.txt
python3 tools/rocpd_gaps.py gpurun_out/final/final_results.db > gpurun_out/final_gaps.txt
bash tools/pmc_run.sh pmc_f FETCH_SIZE -- tools/one_search.py
bash tools/pmc_run.sh pmc_w WRITE_SIZE -- tools/one_search.py
python3 tools/rocpd_summary.py gpurun_out/pmc_f/pmc_f_results.db > gpurun_out/pmc_f.txt
python3 tools/rocpd_summary.py gpurun_out/pmc_w/pmc_w_results.db > gpurun_out/pmc_w.txt
python3 tools/pmc_to_json.py gpurun_out/traffic.json gpurun_out/pmc_f/pmc_f_results.db gpurun_out/pmc_w/pmc_w_results.db
cp gpurun_out/traffic.json profiles/r01_traffic.json      # bench.py reads the traffic figure from profiles/
python3 bench.py > gpurun_out/bench_line.txt 2> gpurun_out/bench_err.txt
tail -c 600 gpurun_out/bench_line.txt
