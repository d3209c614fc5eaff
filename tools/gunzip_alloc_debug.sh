cd $GRAFT_REPO_ROOT
N=1000000 N1=1000000 GZL=1 QUICK=1 timeout 300 bash tools/gz_e2e.sh r06_g2 > /dev/null 2>&1
R="-r tests/golden/data/all_pave_ref.fa.gz -k 16"
for i in 1 2; do RKMH_TIMING=1 RKMH_BGZF_TIMING=1 bin/rkmh stream $R -f /tmp/one.fq.gz -f /tmp/one.fq.gz -f /tmp/one.fq.gz -f /tmp/one.fq.gz 2>&1 >/dev/null | grep "gzip device\|main loop\|slot of" ; echo ---; done
