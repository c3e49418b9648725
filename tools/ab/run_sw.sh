#!/bin/bash
# same-box A/B of the SW pass times
for r in 1 2 3; do
  for v in A B; do
    cp tools/ab/lib$v.so peppan_amd/libpeppan_hip.so
    echo -n "$v "; python bench.py --no-cpu-baseline --steps 20 | grep -o "ms_sw[a-z_]*\": [0-9.]*" | tr '\n' ' '; echo
  done
done
