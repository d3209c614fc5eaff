#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
g++ -O2 -pthread tools/ubench/file_write.cpp -o /tmp/file_write && /tmp/file_write /tmp/fw.bin 2>&1 | tee gpurun_out/r04_file_write.txt
bash tools/c3_variants.sh "-DRK_KMER_SKEW=0" "-DRK_KMER_SKEW=1" "-DRK_KMER_SKEW=0" "-DRK_KMER_SKEW=1" 2>&1 | tee gpurun_out/r04_c3_variants.txt
