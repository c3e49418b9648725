#!/bin/bash
# after `gpurun -- 'bash tools/profile_round.sh'`: copy the summaries from gpurun_out/ into profiles/ with their header lines
set -e
cd "$(dirname "$0")/.."
R=${1:-r03}
{ echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e   (MI355X; tools/rocpd_summary.py over the rocpd database)"; cat gpurun_out/final_stats.txt; } > profiles/${R}_final_kernel_stats.txt
{ echo "# python bench.py  (MI355X; default flags: 1 GPU, 50 steps, 5 warmup, cpu baseline = oracle C port with OpenMP on every host thread, whole workload)"; cat gpurun_out/bench_line.txt; } > profiles/${R}_bench_line.txt
cp gpurun_out/counters.json profiles/${R}_counters.json
{ echo "# tools/micro/valu_rate (MI355X, gfx950): issue rate of the instructions the Smith-Waterman passes are made of; see the header of tools/micro/valu_rate.hip for the method"; cat gpurun_out/valu_rate.txt; } > profiles/${R}_valu_rate.txt
{ echo "# rocprofv3 --kernel-trace --pmc <one pass per line below> -- python3 tools/one_search.py   (10k genes x 1002 nt all-vs-all, 2 searches per pass; MI355X)"
  echo "#   pass 1: FETCH_SIZE   pass 2: WRITE_SIZE   (KiB per dispatch as reported; FETCH_SIZE = TCC_EA0_RDREQ x 64 B: on gfx950 it reads 1/2 of a wide coalesced stream - MI355X_MICROARCH.md section HBM)"
  echo "#   pass 3: SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_INST_ANY   pass 4: GRBM_GUI_ACTIVE (sum over 8 XCDs)"
  for t in pmc_f pmc_w pmc_sq pmc_grbm; do echo "## $t"; sed -n '/^counters/,$p' gpurun_out/$t.txt | tail -n +2; done; } > profiles/${R}_pmc_counters.txt
{ echo "# one step of bench.py on the GPU timeline (tools/rocpd_gaps.py over the same trace as ${R}_final_kernel_stats.txt): start offset, duration, idle gap before each launch"; cat gpurun_out/final_gaps.txt; } > profiles/${R}_step_timeline.txt
