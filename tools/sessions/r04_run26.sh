#!/bin/bash
cd "$(dirname "$0")/../.."
python3 tools/make_fastq.py /tmp/cli1m.fq 16000000
REF=$PWD/tests/golden/data/all_pave_ref.fa.gz
RKMH_RAW=0 python3 -m rkmh_amd.cli stream -r $REF -f /tmp/cli1m.fq -k 16 -s 1000 > /tmp/o.tsv 2> /tmp/e.txt; echo rc=$?; tail -c 1500 /tmp/e.txt; wc -l /tmp/o.tsv
