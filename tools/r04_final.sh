#!/bin/bash
# round-4 closing session on the GPU box: full parity suite, profile of the shipped kernel (kernel stats + PMC), the bench line, the
# workload table, per-kernel stats of the device FASTQ front end, a randomized soak.  Everything lands in gpurun_out/.
cd ${GRAFT_REPO_ROOT:-.}
TAG=${1:-r04_a}
mkdir -p gpurun_out
timeout 2400 python3 -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log
tail -4 gpurun_out/${TAG}_pytest.log
bash tools/profile.sh $TAG > /dev/null 2>&1
cp gpurun_out/$TAG/summary.txt gpurun_out/${TAG}_summary.txt; cp gpurun_out/$TAG/trace/*kernel_stats.csv gpurun_out/${TAG}_kernel_stats.csv 2>/dev/null || cp gpurun_out/$TAG/trace/*/*kernel_stats.csv gpurun_out/${TAG}_kernel_stats.csv
head -12 gpurun_out/${TAG}_summary.txt
python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
python3 -c "
import json; d=json.loads(open('gpurun_out/${TAG}_bench.json').read().strip().split('\n')[-1])
print('value %.4g  kernel_ms %.4f  frac %.4f' % (d['value'], d['roofline']['kernel_ms'], d['roofline']['frac']))
for k in ('cpu_baseline','host_path','depth_filter','c3_panel','c4_filter','c5_call','e2e'):
    v = d.get(k, {}); print(k, {a: v[a] for a in list(v)[:6] if not isinstance(v[a], (list, dict, str)) or len(str(v[a])) < 60})
print('x4', d['e2e'].get('x4'), d['e2e'].get('x4_devnull'))"
timeout 900 python3 tools/bench_configs.py > gpurun_out/${TAG}_configs.txt 2>&1
timeout 300 python3 tools/bench_ragged.py >> gpurun_out/${TAG}_configs.txt 2>&1
cat gpurun_out/${TAG}_configs.txt
python3 tools/make_fastq.py /tmp/prof4m.fq 4000000
cd /tmp; export TMPDIR=/tmp RKMH_SLOW_EXIT=1   # (the binary normally leaves through _exit: the profiler's handlers would not run)
rm -rf /tmp/fqprof; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fqprof -o fq -- $GRAFT_REPO_ROOT/bin/rkmh stream -r $GRAFT_REPO_ROOT/tests/golden/data/all_pave_ref.fa.gz -f /tmp/prof4m.fq -k 16 -s 1000 > /tmp/prof4m.tsv 2> /tmp/prof4m.err
cd $GRAFT_REPO_ROOT; unset RKMH_SLOW_EXIT
cp $(find /tmp/fqprof -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_fastq_kernel_stats.csv
head -14 gpurun_out/${TAG}_fastq_kernel_stats.csv | cut -c1-160
wc -l /tmp/prof4m.tsv
RKMH_TEST_SEEDS=8000 RKMH_TEST_SEED_BASE=1400000 timeout 1200 python3 -m pytest tests/test_gpu_parity.py -q -k "randomized" > gpurun_out/${TAG}_soak1.log 2>&1; echo "rc=$?" >> gpurun_out/${TAG}_soak1.log
tail -2 gpurun_out/${TAG}_soak1.log
RKMH_TEST_SEEDS=1500 RKMH_TEST_SEED_BASE=1500000 RKMH_TEST_LONG=1 timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -k "randomized" > gpurun_out/${TAG}_soak2.log 2>&1; echo "rc=$?" >> gpurun_out/${TAG}_soak2.log
tail -2 gpurun_out/${TAG}_soak2.log
RKMH_TEST_FUZZ=6000 RKMH_TEST_SEED_BASE=99 timeout 900 python3 -m pytest tests/test_gpu_fastq.py -q -k "mutated" > gpurun_out/${TAG}_fuzz1.log 2>&1; echo "rc=$?" >> gpurun_out/${TAG}_fuzz1.log
tail -2 gpurun_out/${TAG}_fuzz1.log
