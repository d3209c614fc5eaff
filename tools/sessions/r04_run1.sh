#!/bin/bash
# round-4 GPU session 1: parity suite on the new k_classify_kmer, then timing + VALU counts of each change
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r04_t1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_t1.log
tail -3 gpurun_out/r04_t1.log
KMER_PMC=1 bash tools/kmer_variants.sh "" "-DRK_KMER_OPT=0 -DRK_KF4_SDWA=0" "-DRK_KMER_OPT=1 -DRK_KF4_SDWA=0" "-DRK_KMER_OPT=2 -DRK_KF4_SDWA=0" "-DRK_KMER_OPT=0 -DRK_KF4_SDWA=1" "-DRK_KMER_OPT=56 -DRK_KF4_SDWA=0" > gpurun_out/r04_variants1.txt 2>&1
cat gpurun_out/r04_variants1.txt
