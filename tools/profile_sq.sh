#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash tools/profile_sq.sh'): VALU issue-rate probe + SQ counters of every kernel of one search.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 tools/micro/valu_rate > gpurun_out/valu_rate.txt 2>&1
bash tools/pmc_run.sh pmc_sq SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_INST_ANY -- tools/one_search.py
python3 tools/rocpd_summary.py gpurun_out/pmc_sq/pmc_sq_results.db > gpurun_out/pmc_sq.txt
bash tools/pmc_run.sh pmc_grbm GRBM_GUI_ACTIVE GRBM_COUNT -- tools/one_search.py
python3 tools/rocpd_summary.py gpurun_out/pmc_grbm/pmc_grbm_results.db > gpurun_out/pmc_grbm.txt
cat gpurun_out/valu_rate.txt
