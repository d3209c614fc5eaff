#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
KMER_PMC=1 bash tools/kmer_variants.sh "-DRK_KMER_SHORT=0" "-DRK_KMER_SHORT=3" "-DRK_KMER_SHORT=4" "-DRK_KMER_SHORT=8" "-DRK_KMER_SHORT=16" 2>&1 | tee gpurun_out/r04_short_c2.txt
bash tools/c3_variants.sh "-DRK_KMER_SHORT=0" "-DRK_KMER_SHORT=4" "-DRK_KMER_SHORT=8" "-DRK_KMER_SHORT=16" 2>&1 | tee gpurun_out/r04_short_c3.txt
