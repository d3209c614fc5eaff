#!/bin/bash
# Usage (on the GPU box): bash tools/kf4_density.sh   -- the k-mer-space kernel against the filter's entries per 16-byte sector
# (RKMH_KF4_ENTRIES, any real number: the sector count need not be a power of two): C2 and synthetic panels of several sizes
for E in 12 12.5 13 13.5 14.5; do
  RKMH_KF4_ENTRIES=$E python3 tools/bench_multik.py 16 2>/dev/null | tail -1
done
for R in 300 400 600 1000 2000; do for E in 14 16 18 20 24; do
  RKMH_KF4_ENTRIES=$E python3 tools/bench_panel.py $R 2>/dev/null | tail -1 | sed "s/^/entries=$E /"
done; done
