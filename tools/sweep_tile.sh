for K in 24 20; do for L in 50 75 100 150 250; do for T in auto 2 4 6; do
  if [ $T = auto ]; then unset RKMH_TILE_T; else export RKMH_TILE_T=$T; fi
  BENCH_K=$K timeout 100 python3 tools/bench_len.py $L 2>/dev/null | tail -1 | sed "s/^/k=$K /"
done; done; done
