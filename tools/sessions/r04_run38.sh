#!/bin/bash
# the GPU suite under forced alternative code paths, second part (after the harness fix): no k-mer-space form; two slots / one writer / small blocks
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
run() { echo "== $*"; env "$@" timeout 1500 python3 -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED" | tail -12; }
run RKMH_KMER_PREFILTER=0
run RKMH_RAW_SLOTS=2 RKMH_OUT_DIRECT=0 RKMH_RAW_BLOCK_KB=512
