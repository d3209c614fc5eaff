cd $GRAFT_REPO_ROOT
[ -f /tmp/one.fq.gz ] || { N=2000000 GZL=6 timeout 400 bash tools/gunzip_profile.sh tmp > /dev/null 2>&1; }
R="-r tests/golden/data/all_pave_ref.fa.gz -k 16"
for ck in 32 16 8; do
  for rep in 1 2; do sleep 1; RKMH_GZIP_CHUNK_KB=$ck RKMH_TIMING=1 RKMH_BGZF_TIMING=1 bin/rkmh stream $R -f /tmp/one.fq.gz -f /tmp/one.fq.gz -f /tmp/one.fq.gz -f /tmp/one.fq.gz 2>&1 >/dev/null | grep "call 1 of\|main loop" | sed "s/^/chunk $ck KB: /" | cut -c1-220; done
done
