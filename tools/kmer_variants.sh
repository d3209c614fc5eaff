#!/bin/bash
# Usage (on the GPU box): bash tools/kmer_variants.sh "<flags of variant 1>" "<flags of variant 2>" ...
# Rebuilds rk_kmer.o (k = 16 kernels only) with each flag set, relinks the library into a scratch copy and prints bench.py's kernel time.
# Flags of the form ENV:NAME=VALUE are exported to the bench run instead of being passed to hipcc.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
cp rkmh_amd/lib/librkmh_amd.so /tmp/librkmh_amd.orig.so
cp rkmh_amd/csrc/rk_kmer.o /tmp/rk_kmer.orig.o
cp rkmh_amd/csrc/rk_api.o /tmp/rk_api.orig.o
for v in "$@"; do
  cf=""; envs=""
  for w in $v; do case $w in ENV:*) envs="$envs ${w#ENV:}";; *) cf="$cf $w";; esac; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DRK_KMER_FAST_BUILD $cf -c rkmh_amd/csrc/rk_kmer.hip -o rkmh_amd/csrc/rk_kmer.o 2>&1 | grep -i "error" 
  # the filter's bit layout (RK_KF4_*) is shared with the host-side builder in rk_api.hip
  case "$cf" in *RK_KF4*|*RK_KMER_INLINE*) /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $cf -c rkmh_amd/csrc/rk_api.hip -o rkmh_amd/csrc/rk_api.o 2>&1 | grep -i "error";; *) cp /tmp/rk_api.orig.o rkmh_amd/csrc/rk_api.o;; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o rkmh_amd/lib/librkmh_amd.so rkmh_amd/csrc/rk_kernels.o rkmh_amd/csrc/rk_classify.o rkmh_amd/csrc/rk_kmer.o rkmh_amd/csrc/rk_count.o rkmh_amd/csrc/rk_call.o rkmh_amd/csrc/rk_sort.o rkmh_amd/csrc/rk_fastq.o rkmh_amd/csrc/rk_fasta.o rkmh_amd/csrc/rk_api.o rkmh_amd/csrc/rk_parse.o rkmh_amd/csrc/rk_format.o rkmh_amd/csrc/rk_synth.o -lz -lpthread
  r=$(env $envs python3 bench.py --steps 100 --warmup 20 --cpu-seconds 0 --no-host-path --no-depth-filter --no-configs --e2e-reads 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%.4f ms' % d['roofline']['kernel_ms'])")
  valu=""
  if [ -n "$KMER_PMC" ]; then
    rm -rf /tmp/kv_pmc; (cd /tmp && TMPDIR=/tmp env $envs rocprofv3 --pmc ${KMER_PMC_COUNTERS:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS} --kernel-trace --output-format csv -d /tmp/kv_pmc -o pmc -- python3 $ROOT/bench.py --steps 10 --warmup 2 --cpu-seconds 0 --no-host-path --no-depth-filter --no-configs --e2e-reads 0 > /dev/null 2>&1)
    valu=$(python3 - <<'PY'
import csv, glob
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob("/tmp/kv_pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "classify" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
print(" ".join("%s=%.2f/read" % (k.replace("SQ_INSTS_", ""), sum(v) / len(v) / 1e6) for k, v in sorted(acc.items())))
PY
)
  fi
  echo "variant [$v]: $r $valu"
done
cp /tmp/librkmh_amd.orig.so rkmh_amd/lib/librkmh_amd.so
cp /tmp/rk_kmer.orig.o rkmh_amd/csrc/rk_kmer.o
cp /tmp/rk_api.orig.o rkmh_amd/csrc/rk_api.o
