#!/bin/bash
# Usage (GPU box): bash tools/inflate_debug.sh -- rebuilds rk_inflate.o with -DRK_INFLATE_DEBUG into a scratch copy of the library and prints the
# per-wave counters of pass 1 (periods, symbol steps, lanes per step, header iterations) for a few launches
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p /tmp/dbg/lib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DRK_INFLATE_DEBUG -c rkmh_amd/csrc/rk_inflate.hip -o /tmp/dbg/rk_inflate.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/dbg/lib/librkmh_amd.so $(ls rkmh_amd/csrc/*.o | grep -v rk_inflate.o) /tmp/dbg/rk_inflate.o -lz -lpthread -ldl || exit 1
[ -f /tmp/big.fq.gz ] || N=2000000 QUICK=1 bash tools/gz_e2e.sh dbg > /dev/null 2>&1
LD_LIBRARY_PATH=/tmp/dbg/lib RKMH_BGZF_DEVICE_WORKERS=1 RKMH_FORK=0 RKMH_SLOW_EXIT=1 LD_PRELOAD=/tmp/dbg/lib/librkmh_amd.so $ROOT/bin/rkmh stream -r $ROOT/tests/golden/data/all_pave_ref.fa.gz -f /tmp/big.fq.gz -k 16 2>&1 >/dev/null | grep "inflate dbg\|place dbg" | head -8
LD_LIBRARY_PATH=/tmp/dbg/lib RKMH_BGZF_DEVICE_WORKERS=1 RKMH_FORK=0 RKMH_SLOW_EXIT=1 LD_PRELOAD=/tmp/dbg/lib/librkmh_amd.so $ROOT/bin/rkmh stream -r $ROOT/tests/golden/data/all_pave_ref.fa.gz -f /tmp/big.fq.gz -k 16 2>/dev/null | grep "inflate dbg\|place dbg" | head -8
