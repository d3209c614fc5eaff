#!/bin/bash
# Usage (GPU box): [CFGS="block_kb workers slots;..."] bash tools/gz_marginal.sh -- device inflate of /tmp/sw.fq.gz (8 M reads, made by tools/gz_sweep.sh):
# wall time of one -f and of three, and the marginal rate (the extra 16 M reads over the extra seconds)
cd ${GRAFT_REPO_ROOT:-.}
[ -f /tmp/sw.fq.gz ] || CFGS="65536 8" bash tools/gz_sweep.sh > /dev/null 2>&1
IFS=";" read -ra LIST <<< "${CFGS:-65536 8 1;65536 8 2;65536 12 1;32768 12 1;32768 16 1;131072 8 1}"
R="-r tests/golden/data/all_pave_ref.fa.gz -k 16"
for cfg in "${LIST[@]}"; do
  set -- $cfg
  export RKMH_BGZF_DEVICE=${DEV:-1} RKMH_RAW_BLOCK_KB=$1 RKMH_RAW_WORKERS=$2 RKMH_RAW_SLOTS=$3
  best1=99; best3=99
  for rep in 1 2; do
    S=$(date +%s.%N); bin/rkmh stream $R -f /tmp/sw.fq.gz > /tmp/sw.out 2>/dev/null; E=$(date +%s.%N)
    best1=$(python3 -c "print(min($best1, $E - $S))")
    S=$(date +%s.%N); bin/rkmh stream $R -f /tmp/sw.fq.gz -f /tmp/sw.fq.gz -f /tmp/sw.fq.gz > /tmp/sw.out 2>/dev/null; E=$(date +%s.%N)
    best3=$(python3 -c "print(min($best3, $E - $S))")
  done
  python3 -c "print('device=%s block_kb=%-7s workers=%-3s slots=%s: 8 M reads %.3f s, 24 M reads %.3f s, marginal %.1f M reads/s' % ('${DEV:-1}', '$1', '$2', '$3', $best1, $best3, 16.0 / ($best3 - $best1)))"
done
