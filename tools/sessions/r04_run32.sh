#!/bin/bash
# what the second probes of the exact map cost (ABL 16: no hop), against the shipped build
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
KMER_PMC=1 KMER_PMC_COUNTERS="SQ_INSTS_VALU TCC_REQ_sum" bash tools/kmer_variants.sh "" "-DRK_KMER_ABL=16" "" "-DRK_KMER_ABL=16" 2>&1 | tee gpurun_out/r04_hops.txt
