#!/bin/bash
# Usage (GPU box): [ONLY="label substring"] bash tools/gz_hybrid.sh -- BGZF (8 M reads, /tmp/sw.fq.gz from tools/gz_sweep.sh) through bin/rkmh stream: host inflate,
# device inflate, both; one -f and three, marginal rate = the extra 16 M reads over the extra seconds
cd ${GRAFT_REPO_ROOT:-.}
[ -f /tmp/sw.fq.gz ] || CFGS="65536 8" timeout 300 bash tools/gz_sweep.sh > /dev/null 2>&1
R="-r tests/golden/data/all_pave_ref.fa.gz -k 16"
run() { # label, env...
  local label=$1; shift
  best1=99; best3=99
  for rep in 1 2 3; do
    S=$(date +%s.%N); env "$@" timeout -s ABRT 60 bin/rkmh stream $R -f /tmp/sw.fq.gz > /tmp/sw.out1 2>/tmp/sw.err || { echo "$label: one file: failed or timed out"; tail -5 /tmp/sw.err; return; }; E=$(date +%s.%N)
    best1=$(python3 -c "print(min($best1, $E - $S))")
    S=$(date +%s.%N); env "$@" timeout -s ABRT 90 bin/rkmh stream $R -f /tmp/sw.fq.gz -f /tmp/sw.fq.gz -f /tmp/sw.fq.gz > /tmp/sw.out 2>/tmp/sw.err || { echo "$label: three files: failed or timed out"; tail -5 /tmp/sw.err; return; }; E=$(date +%s.%N)
    best3=$(python3 -c "print(min($best3, $E - $S))")
  done
  python3 -c "print('%-44s 8 M reads %.3f s, 24 M reads %.3f s, marginal %.1f M reads/s   sha %s' % ('$label', $best1, $best3, 16.0 / ($best3 - $best1), '$(sha256sum /tmp/sw.out1 | cut -c1-12)'))"
}
run "host inflate" RKMH_BGZF_DEVICE=0
run "device inflate (8 workers x 4 jobs)" RKMH_BGZF_DEVICE=1
run "device inflate (12 workers x 4 jobs)" RKMH_BGZF_DEVICE=1 RKMH_BGZF_DEVICE_WORKERS=12
run "both: 6 device workers x 4 jobs" RKMH_BGZF_DEVICE=2
run "both: 8 x 4, 12 host workers" RKMH_BGZF_DEVICE=2 RKMH_RAW_WORKERS=12 RKMH_BGZF_DEVICE_WORKERS=8
run "by size (unset)" RKMH_NOTHING=1
