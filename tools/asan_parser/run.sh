#!/bin/bash
# AddressSanitizer + UBSan over the FASTA/FASTQ front end (host code, no GPU needed): a fuzz corpus of mutated files
# through the block-parallel and the sequential scanner, whose outputs must agree.  Usage: bash tools/asan_parser/run.sh
set -e
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$(cd "$HERE/../.." && pwd); W=${TMPDIR:-/tmp}/rk_asan_parser; mkdir -p $W
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -I$ROOT/include $HERE/main.cpp $ROOT/rkmh_amd/csrc/rk_parse.cpp -o $W/parse_asan -lz -lpthread -ldl
python3 $HERE/gen.py ${1:-1} $W/corpus ${2:-300}
for cfg in "4 4" "8 16" "1 64"; do set -- $cfg
  RKMH_PARSE_THREADS=$1 RKMH_PARSE_BLOCK_KB=$2 $W/parse_asan $W/corpus/*.txt > $W/out_$1.txt 2> $W/err_$1.txt
  echo "threads=$1 block_kb=$2: sanitizer output $(wc -c < $W/err_$1.txt) bytes"
done
cmp $W/out_4.txt $W/out_1.txt && cmp $W/out_8.txt $W/out_1.txt && echo "parallel == sequential on the whole corpus"
