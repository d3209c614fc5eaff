#!/bin/bash
# per-kernel rocprofv3 table of BASELINE config 4 at full size (bin/rkmh filter -k 20 -s 2000 -M 2: reference FASTA stripped on the
# device, reads through the device FASTQ front end, both -M passes).  Inputs come from tools/bench_filter.py (left in /tmp).
cd "$(dirname "$0")/.."
ROOT=$PWD
mkdir -p gpurun_out
python3 tools/bench_filter.py 3100 10000000 > gpurun_out/r04_c4_profile_inputs.txt 2>&1
tail -3 gpurun_out/r04_c4_profile_inputs.txt
cd /tmp; export TMPDIR=/tmp RKMH_SLOW_EXIT=1 RKMH_TIMING=1
for mode in plain M2; do
  extra=""; [ $mode = M2 ] && extra="-M 2"
  rm -rf /tmp/c4prof
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c4prof -o c4 -- $ROOT/bin/rkmh filter -r /tmp/genome_3100.fa -f /tmp/mixed_10000000.fq -k 20 -s 2000 $extra > /tmp/c4prof.out 2> /tmp/c4prof.err
  grep "rkmh timing" /tmp/c4prof.err | head -20
  cp $(find /tmp/c4prof -name "*kernel_stats.csv" | head -1) $ROOT/gpurun_out/r04_c4_${mode}_kernel_stats.csv
  cut -c1-110 $ROOT/gpurun_out/r04_c4_${mode}_kernel_stats.csv | head -24
done
