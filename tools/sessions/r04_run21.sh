#!/bin/bash
# references through the device: tests, then BASELINE config 4 at full size again
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_fasta.py tests/test_gpu_fastq.py -x -q -m gpu 2>&1 | tail -25
echo "pytest rc=$?"
free -g | head -2
timeout 1500 python3 tools/bench_filter.py 3100 10000000 2>&1 | tee gpurun_out/r04_c4_fullsize_c.txt
