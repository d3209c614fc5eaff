cd $GRAFT_REPO_ROOT
[ -f /tmp/sw.fq.gz ] || CFGS="65536 8" timeout 300 bash tools/gz_sweep.sh > /dev/null 2>&1
R="-r tests/golden/data/all_pave_ref.fa.gz -k 16"
for w in 8 12 16; do
  RKMH_BGZF_TIMING=1 RKMH_TIMING=1 RKMH_BGZF_DEVICE=1 RKMH_BGZF_DEVICE_WORKERS=$w timeout -s ABRT 60 bin/rkmh stream $R -f /tmp/sw.fq.gz -f /tmp/sw.fq.gz -f /tmp/sw.fq.gz > /tmp/o.txt 2> /tmp/e.txt
  echo "== $w device workers"; grep "main loop" /tmp/e.txt
  grep "bgzf device" /tmp/e.txt | awk '{r+=$13; c+=$15; e+=$17; w+=$19; n++} END {print n, "device jobs: mean reserve", r/n, "copy", c/n, "enqueue", e/n, "wait", w/n}'
done
