#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
# the probe kernel is not part of the product library: rebuild with it for this measurement
make -s -C peppan_amd/csrc clean && make -s -j8 -C peppan_amd/csrc PROBES=1
for n in 10000 50000; do
  rocprofv3 --kernel-trace --stats -d gpurun_out/pp -o pp -- python3 tools/partition_probe.py $n > gpurun_out/pp_$n.log 2>&1
  python3 tools/rocpd_summary.py gpurun_out/pp/pp_results.db > gpurun_out/pp_${n}_stats.txt
  tail -1 gpurun_out/pp_$n.log; grep -E "kernel  |tgt_slab_probe|seed_match|idx_slab|seed_runs|seed_extend" gpurun_out/pp_${n}_stats.txt
  rm -rf gpurun_out/pp
done
