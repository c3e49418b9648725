#!/bin/bash
# same-box A/B of two builds of the library: tools/ab/libA.so vs tools/ab/libB.so, alternating, N rounds
for r in 1 2 3; do
  for v in A B; do
    cp tools/ab/lib$v.so peppan_amd/libpeppan_hip.so
    echo -n "$v "; python bench.py --no-cpu-baseline --steps 20 | grep -o "ms_per_step\": [0-9.]*"
  done
done
