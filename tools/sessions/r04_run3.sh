#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests -m gpu -x -q > gpurun_out/r04_t3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_t3.log
tail -25 gpurun_out/r04_t3.log
KMER_PMC=1 bash tools/kmer_variants.sh "" "-DRK_KMER_QSTEP=0" "-DRK_KF4_MID=0" "-DRK_KF4_MID=0 -DRK_KMER_QSTEP=0" "" > gpurun_out/r04_variants2.txt 2>&1
cat gpurun_out/r04_variants2.txt
