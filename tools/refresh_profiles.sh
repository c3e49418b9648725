#!/bin/bash
# after `gpurun -- 'bash tools/profile_round.sh'`: copy the summaries from gpurun_out/ into profiles/ with their header lines
set -e
cd "$(dirname "$0")/.."
R=${1:-r06}
{ echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e   (MI355X; tools/rocpd_summary.py over the rocpd database)"; cat gpurun_out/final_stats.txt; } > profiles/${R}_final_kernel_stats.txt
{ echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --genes 50000 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e   (the 50 000 x 50 000 gene all-vs-all of BASELINE configs[4]; MI355X)"; cat gpurun_out/final50_stats.txt; } > profiles/${R}_kernel_stats_50k.txt
{ echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --workload map --map-genomes 16 --steps 2 --warmup 1   (10 000 exemplars x 16 genomes per step, 3 steps: the kernels of the mapping path;"; echo "# GPU-busy fraction of a step = total_us of all kernels / 3 steps / the step's wall time in the line below)"; tail -1 gpurun_out/mapk.log | cut -c1-600; cat gpurun_out/mapk_stats.txt; } > profiles/${R}_map_kernel_stats.txt
{ echo "# python bench.py  (MI355X; default flags: 1 GPU, 250 steps, 5 warmup; cpu baseline = probe for the reference's binaries, then the oracle C port with OpenMP on every host thread)"; cat gpurun_out/bench_line.txt; } > profiles/${R}_bench_line.txt
cp gpurun_out/bench_detail.json profiles/${R}_bench_detail.json      # the full record behind the compact line (bench.py emit)
cp gpurun_out/counters.json profiles/${R}_counters.json
cp gpurun_out/counters_50k.json profiles/${R}_counters_50k.json
cp gpurun_out/counters_blastn.json profiles/${R}_counters_blastn.json
cp gpurun_out/counters_map50k.json profiles/${R}_counters_map50k.json
{ echo "# tools/micro/valu_rate (MI355X, gfx950): issue rate of the instructions the Smith-Waterman passes are made of; see the header of tools/micro/valu_rate.hip for the method"; cat gpurun_out/valu_rate.txt; } > profiles/${R}_valu_rate.txt
{ echo "# tools/sensitive_cost.py 10000 50000 (MI355X): the translated search with DIAMOND's two default seed shapes and with four (pep_set_sensitivity 1)"; grep genes gpurun_out/sensitive_cost.txt; } > profiles/${R}_sensitive_cost.txt
for s in "" _50k _blastn _map50k; do
{ case "$s" in "") w="tools/one_search.py 10000";; _50k) w="tools/one_search.py 50000";; _blastn) w="tools/one_search.py 10000 blastn";; _map50k) w="tools/one_map_step.py 2";; esac
  echo "# rocprofv3 --kernel-trace --pmc <one pass per line below> -- python3 $w  (2 searches / mapping steps per pass; MI355X)"
  echo "#   pass 1: FETCH_SIZE   pass 2: WRITE_SIZE   (KiB per dispatch as reported; FETCH_SIZE = TCC_EA0_RDREQ x 64 B: on gfx950 it reads 1/2 of a wide coalesced stream - MI355X_MICROARCH.md section HBM)"
  echo "#   pass 3: SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_INST_ANY   pass 4: GRBM_GUI_ACTIVE (sum over 8 XCDs)   pass 5: TCC_HIT / MISS / REQ / EA0_RDREQ (sums over the channels)"
  for t in pmc_f pmc_w pmc_sq pmc_grbm pmc_tcc; do echo "## $t$s"; sed -n '/^counters/,$p' gpurun_out/$t$s.txt | tail -n +2; done; } > profiles/${R}_pmc_counters$s.txt
done
{ echo "# one step of bench.py on the GPU timeline (tools/rocpd_gaps.py over the same trace as ${R}_final_kernel_stats.txt): start offset, duration, idle gap before each launch"; cat gpurun_out/final_gaps.txt; } > profiles/${R}_step_timeline.txt
{ echo "# tools/map_pool_rate.py 512 4 8 (MI355X box: 256 hardware threads visible, 16 CPUs granted): get_map_bsn with worker processes, 10 000 exemplars x 512 genomes of 2.2 Mb, four stores written; second pass of a started pool"; cat gpurun_out/map_pool_rate.txt; } > profiles/${R}_map_pool_rate.txt
{ echo "# tools/ab/concurrent_searches.py 10000 200 (MI355X): T host threads with a context (stream, work space) each run the 10 000-gene all-vs-all (log-normal lengths) side by side"; cat gpurun_out/concurrent_searches.txt; } > profiles/${R}_concurrent_searches.txt
