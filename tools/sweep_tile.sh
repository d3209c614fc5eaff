#!/bin/bash
# Usage (on the GPU box): bash tools/sweep_tile.sh
# The hash-space fused kernel (k = 20: compile-time k, k = 24: run-time k) at five read lengths with the automatic choice of reads per
# tile and with 2 / 4 / 6 forced (RKMH_TILE_T): checks make_geom's choice against the best forced tile (profiles/r03_k20_ablation.txt).
for K in 24 20; do for L in 50 75 100 150 250; do for T in auto 2 4 6; do
  if [ $T = auto ]; then unset RKMH_TILE_T; else export RKMH_TILE_T=$T; fi
  BENCH_K=$K timeout 100 python3 tools/bench_len.py $L 2>/dev/null | tail -1 | sed "s/^/k=$K /"
done; done; done
