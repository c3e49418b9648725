#!/bin/bash
# after a gpurun of the profiling commands (see the header lines below) copy the summaries into profiles/ with their headers
set -e
cd "$(dirname "$0")/.."
R=${1:-r01}
{ echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline   (MI355X, round final; tools/rocpd_summary.py over the rocpd database)"; cat gpurun_out/final_stats.txt; } > profiles/${R}_final_kernel_stats.txt
{ echo "# python bench.py  (MI355X; default flags: 1 GPU, 10 steps, 2 warmup, cpu baseline = oracle C port with OpenMP on every host thread, whole workload)"; cat gpurun_out/bench_line.txt; } > profiles/${R}_bench_line.txt
cp gpurun_out/traffic.json profiles/${R}_traffic.json
{ echo "# rocprofv3 --kernel-trace --pmc FETCH_SIZE (pass 1) / WRITE_SIZE (pass 2) -- python3 tools/one_search.py   (10k genes x 1002 nt all-vs-all, 2 searches; MI355X)"; echo "# values: per-dispatch average, unit KiB as reported (FETCH_SIZE = TCC_EA0_RDREQ x 64 B; on gfx950 it reads 1/2 of a wide coalesced stream - MI355X_MICROARCH.md section HBM)"; sed -n '/^counters/,$p' gpurun_out/pmc_f.txt | tail -n +2; sed -n '/^counters/,$p' gpurun_out/pmc_w.txt | tail -n +2; } > profiles/${R}_pmc_hbm_traffic.txt
{ echo "# one step of bench.py on the GPU timeline (tools/rocpd_gaps.py over the same trace as ${R}_final_kernel_stats.txt): start offset, duration, idle gap before each launch"; cat gpurun_out/final_gaps.txt; } > profiles/${R}_step_timeline.txt
