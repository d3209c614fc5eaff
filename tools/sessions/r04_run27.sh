#!/bin/bash
# how much of the drain is the L2 requests of the map look-ups: look-ups without counting (ABL 2) against the same from a 4 KB corner of the map (ABL 258)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
KMER_PMC=1 KMER_PMC_COUNTERS="SQ_INSTS_VALU TCC_REQ_sum TCC_HIT_sum" bash tools/kmer_variants.sh "" "-DRK_KMER_ABL=4" "-DRK_KMER_ABL=2" "-DRK_KMER_ABL=258" "-DRK_KMER_ABL=256" 2>&1 | tee gpurun_out/r04_map_l1.txt
