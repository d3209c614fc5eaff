#!/bin/bash
# where the counting part of the drain goes: no hit multiset (ABL 512), no posting lists (ABL 1024), both
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
KMER_PMC=1 KMER_PMC_COUNTERS="SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" bash tools/kmer_variants.sh "" "-DRK_KMER_ABL=512" "-DRK_KMER_ABL=1024" "-DRK_KMER_ABL=1536" "-DRK_KMER_ABL=2" 2>&1 | tee gpurun_out/r04_counting_split.txt
