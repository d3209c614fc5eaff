#!/bin/bash
# what 1/3 fewer filter requests would buy (ABL 2048), alone and without the drain (ABL 2052 against ABL 4); the paired form (128) beside it
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
KMER_PMC=1 KMER_PMC_COUNTERS="SQ_INSTS_VALU TCC_REQ_sum" bash tools/kmer_variants.sh "" "-DRK_KMER_ABL=2048" "-DRK_KMER_ABL=128" "-DRK_KMER_ABL=4" "-DRK_KMER_ABL=2052" "-DRK_KMER_ABL=132" "" 2>&1 | tee gpurun_out/r04_g6_proxy.txt
