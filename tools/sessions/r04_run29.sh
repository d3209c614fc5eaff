#!/bin/bash
# posting lists of deferred hits: dependent loads (0), length + postings together (1), software-pipelined over the hits (2)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
KMER_PMC=1 KMER_PMC_COUNTERS="SQ_INSTS_VALU SQ_INSTS_LDS" bash tools/kmer_variants.sh "-DRK_KMER_MQ_MODE=0" "-DRK_KMER_MQ_MODE=1" "-DRK_KMER_MQ_MODE=2" "-DRK_KMER_MQ_MODE=0" 2>&1 | tee gpurun_out/r04_mq_mode.txt
bash tools/c3_variants.sh "-DRK_KMER_MQ_MODE=0" "-DRK_KMER_MQ_MODE=1" "-DRK_KMER_MQ_MODE=2" 2>&1 | tee -a gpurun_out/r04_mq_mode.txt
