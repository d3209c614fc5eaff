#!/bin/bash
# Usage (on the GPU box): K=20 bash tools/classify_variants.sh "<flags of variant 1>" "<flags of variant 2>" ...
# Rebuilds rk_classify.o for ONE compile-time k (-DRK_TILE_ONLY_K=$K) with each flag set, relinks the library and prints
# tools/bench_len.py's time for that k plus the instruction counters of the kernel.  ENV:NAME=VALUE entries go to the run's environment.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
K=${K:-20}; L=${L:-150}
cp rkmh_amd/lib/librkmh_amd.so /tmp/librkmh_amd.orig.so
cp rkmh_amd/csrc/rk_classify.o /tmp/rk_classify.orig.o
for v in "$@"; do
  cf=""; envs=""
  for w in $v; do case $w in ENV:*) envs="$envs ${w#ENV:}";; *) cf="$cf $w";; esac; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DRK_TILE_ONLY_K=$K $cf -c rkmh_amd/csrc/rk_classify.hip -o rkmh_amd/csrc/rk_classify.o 2>&1 | grep -i "error"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o rkmh_amd/lib/librkmh_amd.so rkmh_amd/csrc/rk_kernels.o rkmh_amd/csrc/rk_classify.o rkmh_amd/csrc/rk_kmer.o rkmh_amd/csrc/rk_count.o rkmh_amd/csrc/rk_call.o rkmh_amd/csrc/rk_api.o rkmh_amd/csrc/rk_parse.o rkmh_amd/csrc/rk_synth.o -lz -lpthread
  r=$(env BENCH_K=$K $envs python3 tools/bench_len.py $L 2>/dev/null | tail -1 | sed 's/.*: //')
  rm -rf /tmp/cv_pmc
  (cd /tmp && TMPDIR=/tmp env BENCH_K=$K $envs rocprofv3 --pmc ${PMC_COUNTERS:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS} --kernel-trace --output-format csv -d /tmp/cv_pmc -o pmc -- python3 $ROOT/tools/bench_len.py $L > /dev/null 2>&1)
  c=$(python3 - <<'PY'
import csv, glob
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob("/tmp/cv_pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "classify" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
print(" ".join("%s=%.1f/read" % (k.replace("SQ_INSTS_", ""), sum(v) / len(v) / 1e6) for k, v in sorted(acc.items())))
PY
)
  echo "variant [$v]: $r $c"
done
cp /tmp/librkmh_amd.orig.so rkmh_amd/lib/librkmh_amd.so
cp /tmp/rk_classify.orig.o rkmh_amd/csrc/rk_classify.o
