#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_fastq.py tests/test_gpu_fasta.py -x -q -m gpu 2>&1 | tail -25
