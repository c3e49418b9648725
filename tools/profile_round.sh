#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash tools/profile_round.sh'): collects everything tools/refresh_profiles.sh copies into profiles/.
#   kernel trace + stats of the default bench, the step timeline, FETCH_SIZE / WRITE_SIZE / SQ / GRBM / TCC counters in separate PMC passes - over the
#   headline workload (10 000 genes), over the 50 000-gene all-vs-all of the `workloads` block, over the nucleotide tool on the headline's genes and
#   over a mapping step of the map_50k leg -, the kernel table of a mapping step, the VALU issue-rate probe, the gather-rate probe, the bench line
#   (which reads profiles/r06_counters*.json written here).
R=r06
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 tools/micro/valu_rate > gpurun_out/valu_rate.txt 2>&1
timeout 200 tools/micro/gather_rate > gpurun_out/gather_rate.txt 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/final -o final -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e > gpurun_out/final.log 2>&1
python3 tools/rocpd_summary.py gpurun_out/final/final_results.db > gpurun_out/final_stats.txt
python3 tools/rocpd_gaps.py gpurun_out/final/final_results.db > gpurun_out/final_gaps.txt
rocprofv3 --kernel-trace --stats -d gpurun_out/final50 -o final50 -- python3 bench.py --genes 50000 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e > gpurun_out/final50.log 2>&1
python3 tools/rocpd_summary.py gpurun_out/final50/final50_results.db > gpurun_out/final50_stats.txt
rocprofv3 --kernel-trace --stats -d gpurun_out/mapk -o mapk -- python3 bench.py --workload map --map-genomes 16 --steps 2 --warmup 1 > gpurun_out/mapk.log 2>&1
python3 tools/rocpd_summary.py gpurun_out/mapk/mapk_results.db > gpurun_out/mapk_stats.txt
pmc_set() {   # pmc_set <suffix> <workload text;steps per pass | n_genes> <python script + args ...>: the five counter passes of one workload -> profiles/${R}_counters<suffix>.json
  local s=$1 w=$2; shift 2
  bash tools/pmc_run.sh pmc_f$s FETCH_SIZE -- "$@"
  bash tools/pmc_run.sh pmc_w$s WRITE_SIZE -- "$@"
  bash tools/pmc_run.sh pmc_sq$s SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_INST_ANY -- "$@"
  bash tools/pmc_run.sh pmc_grbm$s GRBM_GUI_ACTIVE -- "$@"
  bash tools/pmc_run.sh pmc_tcc$s TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -- "$@"
  for t in pmc_f pmc_w pmc_sq pmc_grbm pmc_tcc; do python3 tools/rocpd_summary.py gpurun_out/$t$s/${t}${s}_results.db > gpurun_out/$t$s.txt; done
  python3 tools/pmc_to_json.py gpurun_out/counters$s.json "$w" gpurun_out/pmc_f$s/pmc_f${s}_results.db gpurun_out/pmc_w$s/pmc_w${s}_results.db gpurun_out/pmc_sq$s/pmc_sq${s}_results.db gpurun_out/pmc_grbm$s/pmc_grbm${s}_results.db gpurun_out/pmc_tcc$s/pmc_tcc${s}_results.db
  cp gpurun_out/counters$s.json profiles/${R}_counters$s.json      # bench.py reads the per-kernel counters from profiles/
  rm -rf gpurun_out/pmc_f$s gpurun_out/pmc_w$s gpurun_out/pmc_sq$s gpurun_out/pmc_grbm$s gpurun_out/pmc_tcc$s
}
pmc_set "" 10000 tools/one_search.py 10000
pmc_set _50k 50000 tools/one_search.py 50000
pmc_set _blastn "10000 genes x 1002 nt against themselves, nucleotide tool, both strands (tools/one_search.py 10000 blastn: 2 searches per pass);2" tools/one_search.py 10000 blastn
pmc_set _map50k "50000 exemplar genes x 1002 nt mapped onto 8 genomes of a 50000-gene pan-genome, both tools (tools/one_map_step.py 2: 2 mapping steps per pass);2" tools/one_map_step.py 2
{ echo "# tools/micro/valu_rate (MI355X, gfx950): issue rate of the instructions the Smith-Waterman passes are made of; see the header of tools/micro/valu_rate.hip for the method"; cat gpurun_out/valu_rate.txt; } > profiles/${R}_valu_rate.txt
python3 tools/sensitive_cost.py 10000 50000 > gpurun_out/sensitive_cost.txt 2>&1
python3 tools/map_pool_rate.py 512 4 8 2>&1 | grep '^workers\|container CPU' > gpurun_out/map_pool_rate.txt
{ echo "# tools/map_pool_rate.py 2000 8: BASELINE configs[4]'s genome count at 10 000 exemplars on ONE GPU"; python3 tools/map_pool_rate.py 2000 8 2>&1 | grep '^workers\|container CPU'; } >> gpurun_out/map_pool_rate.txt
{ echo "# GENES=50000 PRESENCE=pan tools/map_pool_rate.py 128 0 4 8: BASELINE configs[4]'s exemplar count - 50 000 genes, genomes of ~6 500 of them (7.7 Mb) -, one process (cold) and 4 / 8 workers"; GENES=50000 PRESENCE=pan python3 tools/map_pool_rate.py 128 0 4 8 2>&1 | grep '^workers\|container CPU'; } >> gpurun_out/map_pool_rate.txt
python3 tools/ab/concurrent_searches.py 10000 200 2>&1 | grep '^threads' > gpurun_out/concurrent_searches.txt
python3 bench.py > gpurun_out/bench_line.txt 2> gpurun_out/bench_err.txt
tail -c 1500 gpurun_out/bench_line.txt
