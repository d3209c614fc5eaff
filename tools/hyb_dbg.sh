cd $GRAFT_REPO_ROOT
[ -f /tmp/sw.fq.gz ] || CFGS="65536 8" timeout 300 bash tools/gz_sweep.sh > /dev/null 2>&1
python3 - <<'PY'
import time, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from rkmh_amd import api
for i in range(3):
    t = time.perf_counter(); z = api.Bgzf.open("/tmp/sw.fq.gz"); dt = time.perf_counter() - t
    print("rk_bgzf_open: %.1f ms, %d members, %.2f GB text" % (dt * 1e3, z.members, z.text_bytes / 1e9)); z.close()
PY
R="-r tests/golden/data/all_pave_ref.fa.gz -k 16"
RKMH_TIMING=1 RKMH_BGZF_DEVICE=1 RKMH_RAW_BLOCK_KB=65536 RKMH_RAW_WORKERS=8 timeout -s ABRT 60 bin/rkmh stream $R -f /tmp/sw.fq.gz -f /tmp/sw.fq.gz -f /tmp/sw.fq.gz 2>&1 > /tmp/o.txt | grep timing
