#!/bin/bash
# filter / -M through the device front end: tests, then BASELINE config 4 at full size again
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_fastq.py -x -q -m gpu 2>&1 | tail -15
echo "pytest rc=$?"


