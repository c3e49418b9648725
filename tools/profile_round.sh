#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash tools/profile_round.sh'): collects everything tools/refresh_profiles.sh copies into profiles/.
#   kernel trace + stats of the default bench, the step timeline, FETCH_SIZE / WRITE_SIZE / SQ / GRBM counters in separate PMC passes,
#   the VALU issue-rate probe, the bench line (which reads profiles/r03_counters.json written here).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 tools/micro/valu_rate > gpurun_out/valu_rate.txt 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/final -o final -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e > gpurun_out/final.log 2>&1
python3 tools/rocpd_summary.py gpurun_out/final/final_results.db > gpurun_out/final_stats.txt
python3 tools/rocpd_gaps.py gpurun_out/final/final_results.db > gpurun_out/final_gaps.txt
bash tools/pmc_run.sh pmc_f FETCH_SIZE -- tools/one_search.py
bash tools/pmc_run.sh pmc_w WRITE_SIZE -- tools/one_search.py
bash tools/pmc_run.sh pmc_sq SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_INST_ANY -- tools/one_search.py
bash tools/pmc_run.sh pmc_grbm GRBM_GUI_ACTIVE -- tools/one_search.py
for t in pmc_f pmc_w pmc_sq pmc_grbm; do python3 tools/rocpd_summary.py gpurun_out/$t/${t}_results.db > gpurun_out/$t.txt; done
python3 tools/pmc_to_json.py gpurun_out/counters.json gpurun_out/pmc_f/pmc_f_results.db gpurun_out/pmc_w/pmc_w_results.db gpurun_out/pmc_sq/pmc_sq_results.db gpurun_out/pmc_grbm/pmc_grbm_results.db
cp gpurun_out/counters.json profiles/r03_counters.json      # bench.py reads the per-kernel counters from profiles/
{ echo "# tools/micro/valu_rate (MI355X, gfx950): issue rate of the instructions the Smith-Waterman passes are made of; see the header of tools/micro/valu_rate.hip for the method"; cat gpurun_out/valu_rate.txt; } > profiles/r03_valu_rate.txt
python3 bench.py > gpurun_out/bench_line.txt 2> gpurun_out/bench_err.txt
tail -c 1500 gpurun_out/bench_line.txt
