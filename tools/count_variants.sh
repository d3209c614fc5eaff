#!/bin/bash
# Usage (on the GPU box): bash tools/count_variants.sh "<flags of variant 1>" ...   (rebuilds rk_count.o, times the slot-partitioned count pass)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
cp rkmh_amd/lib/librkmh_amd.so /tmp/librkmh_amd.orig.so
cp rkmh_amd/csrc/rk_count.o /tmp/rk_count.orig.o
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $v -c rkmh_amd/csrc/rk_count.hip -o rkmh_amd/csrc/rk_count.o 2>&1 | grep -i "error"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o rkmh_amd/lib/librkmh_amd.so rkmh_amd/csrc/rk_kernels.o rkmh_amd/csrc/rk_classify.o rkmh_amd/csrc/rk_kmer.o rkmh_amd/csrc/rk_count.o rkmh_amd/csrc/rk_call.o rkmh_amd/csrc/rk_api.o rkmh_amd/csrc/rk_parse.o rkmh_amd/csrc/rk_synth.o -lz -lpthread
  echo "variant [$v]"
  SLOTS=${SLOTS:-200000000,10000000} FORMS=1 python3 tools/bench_count_forms.py 16 2>/dev/null
  rm -rf /tmp/cv; (cd /tmp && TMPDIR=/tmp SLOTS=200000000 FORMS=1 REPS=5 rocprofv3 --kernel-trace --stats -d /tmp/cv -o cv --output-format csv -- python3 $ROOT/tools/bench_count_forms.py 16 > /dev/null 2>&1)
  python3 - <<'PY'
import csv, glob
for f in glob.glob("/tmp/cv/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(t in r["Name"] for t in ("k_slot", "k_count_bins", "k_classify_tile", "k_max_len", "fillBuffer")):
            print("    %-40s avg %8.1f us" % (r["Name"].split("(")[0][:40], float(r["AverageNs"]) / 1e3))
PY
done
cp /tmp/librkmh_amd.orig.so rkmh_amd/lib/librkmh_amd.so
cp /tmp/rk_count.orig.o rkmh_amd/csrc/rk_count.o
