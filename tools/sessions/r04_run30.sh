#!/bin/bash
# value table with inline lists of 3..6 references: parity first, then A/B at C2 and on the bundled panel
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
RKMH_TEST_SEEDS=1500 timeout 1200 python3 -m pytest tests/test_gpu_parity.py -x -q -k "randomized or golden or edge or kmer_space or c3_sized or full_size" 2>&1 | tail -5
KMER_PMC=1 KMER_PMC_COUNTERS="SQ_INSTS_VALU SQ_INSTS_LDS" bash tools/kmer_variants.sh "-DRK_KMER_INLINE_N=0" "-DRK_KMER_INLINE_N=1" "-DRK_KMER_INLINE_N=0" "-DRK_KMER_INLINE_N=1" 2>&1 | tee gpurun_out/r04_inline_n.txt
bash tools/c3_variants.sh "-DRK_KMER_INLINE_N=0" "-DRK_KMER_INLINE_N=1" 2>&1 | tee -a gpurun_out/r04_inline_n.txt
