#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_fasta.py -x -q -m gpu 2>&1 | tail -25
RKMH_TEST_FUZZ=4000 RKMH_TEST_SEED_BASE=5 timeout 900 python -m pytest tests/test_gpu_fasta.py -x -q -m gpu -k mutated 2>&1 | tail -5
