#!/bin/bash
# the GPU suite under forced alternative code paths (final build)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
run() { echo "== $*"; env "$@" timeout 1500 python3 -m pytest tests -m gpu -q -x 2>&1 | tail -3; }
run RKMH_KMER_PREFILTER=0
run RKMH_RAW_SLOTS=2 RKMH_OUT_DIRECT=0 RKMH_RAW_BLOCK_KB=512
run RKMH_COUNT_BINS=1
run RKMH_COUNT_BINS=0 RKMH_HOST_REGISTER=0
