#!/bin/bash
# counters of the seed-stage kernels at N genes: cache behaviour and issue statistics, one --pmc pass per group
n=${1:-50000}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 -L > gpurun_out/pmc_list.txt 2>&1
grep -oE "\b(TCP|TCC|SQ|TA|TD)_[A-Za-z0-9_]+" gpurun_out/pmc_list.txt | sort -u > gpurun_out/pmc_names.txt
wc -l gpurun_out/pmc_names.txt
pass() { tag=$1; shift; bash tools/pmc_run.sh $tag "$@" -- tools/one_search.py $n; python3 tools/rocpd_summary.py gpurun_out/$tag/${tag}_results.db | sed -n '/counters/,$p' | grep -E "seed_|idx_" ; }
pass ps_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVES
pass ps_sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS
pass ps_tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
pass ps_tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum
