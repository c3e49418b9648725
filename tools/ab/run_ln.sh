#!/bin/bash
# same-box A/B on the log-normal gene lengths workload
for r in 1 2; do
  for v in A B; do
    cp tools/ab/lib$v.so peppan_amd/libpeppan_hip.so
    echo -n "$v "; python bench.py --no-cpu-baseline --steps 10 --gene-len 0 | grep -o "ms_per_step\": [0-9.]*\|ms_sw[a-z_]*\": [0-9.]*" | tr '\n' ' '; echo
  done
done
