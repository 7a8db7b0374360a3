#!/bin/bash
# round 4, first GPU call: GPU tests, the default bench (compact line on stdout, full object in gpurun_out/), per-layer times in the default
# (h3) and the fast (mx6) arithmetic, single stream
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( time python -m pytest tests -m gpu -x -q ) > gpurun_out/r4_gputest.log 2>&1
tail -3 gpurun_out/r4_gputest.log
( time python bench.py ) > gpurun_out/r4_bench_stdout.log 2> gpurun_out/r4_bench_stderr.log
tail -c 5000 gpurun_out/r4_bench_stdout.log | tail -1 > gpurun_out/r4_bench_line.json
wc -c gpurun_out/r4_bench_line.json
cat gpurun_out/r4_bench_line.json
tail -5 gpurun_out/r4_bench_stderr.log
run() { python bench.py --sub --no-overlap --profile-filter '' --steps 10 --warmup 3 --cpu-frames 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
L = d['roofline']['layers']
print('$1', 'fps', d['value'], 'sum_us', round(sum(v['us'] for v in L.values())), ' '.join(f\"{k}={v['us']:.0f}\" for k, v in L.items()))"; }
run h3 | tee -a gpurun_out/r4_layer_times.txt
EVR_ARITH=mx6 run mx6 | tee -a gpurun_out/r4_layer_times.txt
