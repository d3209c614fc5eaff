#!/bin/bash
# the one-process-per-GPU front end (python -m rkmh_amd.cli) end to end on 16 M reads: one file, the same file four times, and two
# ranks on the one GPU (gloo) -- whole-process wall times; the marginal rate is 3 n / (t4 - t1)
cd "$(dirname "$0")/.."
ROOT=$PWD
mkdir -p gpurun_out
python3 tools/make_fastq.py /tmp/cli16m.fq 16000000
REF=$ROOT/tests/golden/data/all_pave_ref.fa.gz
run() { local t0=$(date +%s.%N); "$@" > /tmp/cli.out 2> /tmp/cli.err; local rc=$?; local t1=$(date +%s.%N); echo "rc=$rc wall=$(python3 -c "print('%.2f' % ($t1 - $t0))") s lines=$(wc -l < /tmp/cli.out)"; grep "rkmh timing" /tmp/cli.err | head -4; }
echo "bin/rkmh stream x1"; run bin/rkmh stream -r $REF -f /tmp/cli16m.fq -k 16 -s 1000
echo "cli x1"; RKMH_TIMING=1 run python3 -m rkmh_amd.cli stream -r $REF -f /tmp/cli16m.fq -k 16 -s 1000
echo "cli x4"; RKMH_TIMING=1 run python3 -m rkmh_amd.cli stream -r $REF -f /tmp/cli16m.fq -f /tmp/cli16m.fq -f /tmp/cli16m.fq -f /tmp/cli16m.fq -k 16 -s 1000
echo "cli x1, host parser per rank (RKMH_RAW=0: the path of round 3)"; RKMH_RAW=0 run python3 -m rkmh_amd.cli stream -r $REF -f /tmp/cli16m.fq -k 16 -s 1000
echo "2 ranks on one GPU (gloo) x4"; RKMH_TIMING=1 RKMH_ONE_DEVICE=1 RKMH_DIST_BACKEND=gloo run python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29711 -m rkmh_amd.cli stream -r $REF -f /tmp/cli16m.fq -f /tmp/cli16m.fq -f /tmp/cli16m.fq -f /tmp/cli16m.fq -k 16 -s 1000
