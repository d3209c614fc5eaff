#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
RKMH_TEST_SEEDS=12000 RKMH_TEST_SEED_BASE=1400000 timeout 1500 python3 -m pytest tests/test_gpu_parity.py -q -k "randomized" > gpurun_out/r04_soak1.log 2>&1; echo "rc=$?" >> gpurun_out/r04_soak1.log
tail -3 gpurun_out/r04_soak1.log
RKMH_TEST_SEEDS=2000 RKMH_TEST_SEED_BASE=1500000 RKMH_TEST_LONG=1 timeout 1200 python3 -m pytest tests/test_gpu_parity.py -q -k "randomized" > gpurun_out/r04_soak2.log 2>&1; echo "rc=$?" >> gpurun_out/r04_soak2.log
tail -3 gpurun_out/r04_soak2.log
RKMH_TEST_FUZZ=8000 RKMH_TEST_SEED_BASE=99 timeout 1200 python3 -m pytest tests/test_gpu_fastq.py -q -k "mutated" > gpurun_out/r04_fuzz1.log 2>&1; echo "rc=$?" >> gpurun_out/r04_fuzz1.log
tail -3 gpurun_out/r04_fuzz1.log
