#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench_l2_requests.hip -o /tmp/ubench_l2 2>/dev/null && /tmp/ubench_l2 > gpurun_out/r04_l2_requests.txt 2>&1
cat gpurun_out/r04_l2_requests.txt
KMER_PMC=1 KMER_PMC_COUNTERS="SQ_INSTS_VALU TCP_TCC_READ_REQ_sum" bash tools/kmer_variants.sh "" "-DRK_KMER_ABL=4" "-DRK_KMER_ABL=132" "-DRK_KMER_ABL=128" "-DRK_KMER_ABL=12" "-DRK_KMER_ABL=140" > gpurun_out/r04_variants3.txt 2>&1
cat gpurun_out/r04_variants3.txt
