#!/bin/bash
# long soaks on the final build: randomized differential cases (other seeds than the closing session's), both fuzzers
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
RKMH_TEST_SEEDS=24000 RKMH_TEST_SEED_BASE=2600000 timeout 2400 python3 -m pytest tests/test_gpu_parity.py -q -k "randomized" 2>&1 | tail -2
RKMH_TEST_SEEDS=3000 RKMH_TEST_SEED_BASE=2700000 RKMH_TEST_LONG=1 timeout 1500 python3 -m pytest tests/test_gpu_parity.py -q -k "randomized" 2>&1 | tail -2
RKMH_TEST_FUZZ=20000 RKMH_TEST_SEED_BASE=1234 timeout 1500 python3 -m pytest tests/test_gpu_fastq.py -q -k "mutated" 2>&1 | tail -2
RKMH_TEST_FUZZ=30000 RKMH_TEST_SEED_BASE=4321 timeout 1500 python3 -m pytest tests/test_gpu_fasta.py -q -k "mutated" 2>&1 | tail -2
