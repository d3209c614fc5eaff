#!/bin/bash
# Usage (GPU box): bash tools/c3_variants.sh "<hipcc flags of variant 1>" ...   -- rebuilds rk_kmer.o (k = 16 only) per variant and runs tools/c3_probe.py
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
cp rkmh_amd/lib/librkmh_amd.so /tmp/librkmh_amd.orig.so
cp rkmh_amd/csrc/rk_kmer.o /tmp/rk_kmer.orig.o
cp rkmh_amd/csrc/rk_api.o /tmp/rk_api.orig.o
for v in "$@"; do
  case "$v" in *RK_KF4*|*RK_KMER_INLINE*) /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $v -c rkmh_amd/csrc/rk_api.hip -o rkmh_amd/csrc/rk_api.o 2>&1 | grep -i "error";; *) cp /tmp/rk_api.orig.o rkmh_amd/csrc/rk_api.o;; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DRK_KMER_FAST_BUILD $v -c rkmh_amd/csrc/rk_kmer.hip -o rkmh_amd/csrc/rk_kmer.o 2>&1 | grep -i "error"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o rkmh_amd/lib/librkmh_amd.so rkmh_amd/csrc/rk_kernels.o rkmh_amd/csrc/rk_classify.o rkmh_amd/csrc/rk_kmer.o rkmh_amd/csrc/rk_count.o rkmh_amd/csrc/rk_call.o rkmh_amd/csrc/rk_sort.o rkmh_amd/csrc/rk_fastq.o rkmh_amd/csrc/rk_fasta.o rkmh_amd/csrc/rk_api.o rkmh_amd/csrc/rk_parse.o rkmh_amd/csrc/rk_format.o rkmh_amd/csrc/rk_synth.o -lz -lpthread
  echo "variant [$v]"
  python3 tools/c3_probe.py 2>/dev/null | grep -v "^references"
done
cp /tmp/librkmh_amd.orig.so rkmh_amd/lib/librkmh_amd.so
cp /tmp/rk_kmer.orig.o rkmh_amd/csrc/rk_kmer.o
cp /tmp/rk_api.orig.o rkmh_amd/csrc/rk_api.o
