#!/bin/bash
# seed_runs_extend variants: RUN_TRIP (hits turned into runs per trip) and grid size, same box; usage: bash tools/ab/phase2_ab.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "2 8" "4 8" "2 16" "4 16" "1 8"; do
  set -- $v
  sed -i "s/^constexpr int RUN_TRIP = [0-9]*;/constexpr int RUN_TRIP = $1;/; s/hipLaunchKernelGGL(seed_runs_extend, dim3(256u \* [0-9]*u)/hipLaunchKernelGGL(seed_runs_extend, dim3(256u * $2u)/" peppan_amd/csrc/seeds.hip
  make -s -j8 -C peppan_amd/csrc 2>&1 | grep -E "error" 
  for g in 10000 50000; do
    rocprofv3 --kernel-trace --stats -d gpurun_out/ab -o ab -- python3 bench.py --genes $g --steps 5 --warmup 2 --no-cpu-baseline --no-e2e > gpurun_out/ab.log 2>&1
    echo "RUN_TRIP $1 grid 256x$2 genes $g: $(python3 tools/rocpd_summary.py gpurun_out/ab/ab_results.db | grep seed_runs_extend | awk '{print $4, "us avg"}')"; rm -rf gpurun_out/ab
  done
done
