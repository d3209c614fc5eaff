#!/bin/bash
# Usage: tools/pin_from_mkmh.sh <path to a checkout of github.com/edawson/mkmh> [--apply]
#
# Turns "parity unpinned"  into a ten-minute job the day mkmh's source is available.  mkmh (+ its murmur3
# sub-directory) is the un-vendored submodule of the reference (/root/reference/.gitmodules:1-3; rkmh.cpp:17,21); nothing of it
# exists in this container, so this script CANNOT run here today.  What it does:
#   1. builds oracle/_ref/pin_driver from oracle/pin/pin_driver.cpp + the checkout's mkmh.cpp and murmur3 sources with plain g++
#      (not the reference's Makefile); outputs only under oracle/_ref/ (git-ignored, like every reference-derived build)
#   2. runs it: probes of calc_hash / calc_hashes / to_upper / minhashes / hash_intersection_size / HASHTCounter / mask_by_frequency
#   3. tools/pin_compare.py replays the same probes through the oracle under every candidate policy and prints which of U1-U12
#      hold and which policy constants must flip (oracle/rk_oracle.h rko_policy defaults, rkmh_amd/csrc rk_policy defaults)
#   4. with --apply: rewrites the policy defaults, regenerates tests/golden/*.json with tests/golden/gen_golden.py and runs the
#      CPU test-suite, so the goldens become reference-pinned
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
MK=${1:?usage: tools/pin_from_mkmh.sh <mkmh checkout> [--apply]}
[ -f "$MK/mkmh.hpp" ] || { echo "no mkmh.hpp under $MK" >&2; exit 2; }
mkdir -p "$ROOT/oracle/_ref"
SRCS="$ROOT/oracle/pin/pin_driver.cpp"
[ -f "$MK/mkmh.cpp" ] && SRCS="$SRCS $MK/mkmh.cpp"
for f in "$MK"/murmur3/murmur3.cpp "$MK"/murmur3/murmur3.c "$MK"/murmur3/MurmurHash3.cpp; do [ -f "$f" ] && SRCS="$SRCS $f"; done
INC="-I$MK -I$MK/murmur3"
[ -f "$MK/HASHTCounter.hpp" ] || INC="$INC -I$ROOT/../reference/src -I/root/reference/src"   # rkmh keeps HASHTCounter.hpp in its own src/
set -x
g++ -O2 -std=c++11 -fopenmp $INC $SRCS -o "$ROOT/oracle/_ref/pin_driver"
"$ROOT/oracle/_ref/pin_driver" > "$ROOT/oracle/_ref/pin_probes.json"
set +x
python3 "$ROOT/tools/pin_compare.py" "$ROOT/oracle/_ref/pin_probes.json" ${2:-}
